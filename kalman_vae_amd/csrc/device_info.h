// Properties of the current device that launch decisions depend on.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

namespace kvae {

// Compute units of the CURRENT device, looked up once per device ordinal (a process may
// drive several GPUs, of different sizes).  256 when no device answers.
inline int device_compute_units() {
    constexpr int MAX_DEVICES = 64;
    static std::atomic<int> cached[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int q = 0;
        n = (hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
             q > 0) ? q : 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

}  // namespace kvae
