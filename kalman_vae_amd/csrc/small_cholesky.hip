// Cholesky factor of ONE small symmetric positive-definite matrix (the noise
// covariances Q (n,n) and R (p,p)) and its adjoint, as two single-wavefront
// kernels.  The factor is what the reference's MultivariateNormal builds for its
// reparameterised samples (kvae/state_space_model.py:242-247, 261-266, 409-413).
// A library factorisation works too, but reports failure through a host
// synchronisation and may allocate: neither is allowed inside a HIP graph.
// Failure (non-positive pivot) sets bit 8 of the device status word instead.
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"

namespace kvae {

constexpr int CHOL_MAX_N = 64;      // float; double: 32 (static LDS budget)

// L (lower, zeros above) with L L^T = A.  One wavefront; lane i owns row i.
template <typename T, int MAXN>
__global__ void __launch_bounds__(64)
small_cholesky_forward_kernel(int n, const T* __restrict__ A, T* __restrict__ L, int* status) {
    __shared__ T s[MAXN * (MAXN + 1)];
    const int i = threadIdx.x, ld = n + 1;
    for (int e = i; e < n * n; e += 64) s[(e / n) * ld + e % n] = A[e];
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < n; ++j) {
        // row i of the finished columns 0..j-1 is final for every i >= j
        T acc = T(0);
        if (i >= j && i < n) {
            acc = s[i * ld + j];
            for (int k = 0; k < j; ++k) acc -= s[i * ld + k] * s[j * ld + k];
        }
        __syncthreads();
        if (i == j) {
            if (!(acc > T(0))) bad = true;
            s[j * ld + j] = sqrt(acc);
        }
        __syncthreads();
        if (i > j && i < n) s[i * ld + j] = acc / s[j * ld + j];
        __syncthreads();
    }
    for (int e = i; e < n * n; e += 64) {
        const int r = e / n, c = e % n;
        L[e] = c <= r ? s[r * ld + c] : T(0);
    }
    if (bad && status) atomicOr(status, 8);
}

// gA (symmetric) = adjoint of A -> chol(A) applied to gL (only its lower triangle
// is used).  Reverse sweep over the statements of the column-by-column
// factorisation; lane k owns column k of the running adjoint of L.
template <typename T, int MAXN>
__global__ void __launch_bounds__(64)
small_cholesky_backward_kernel(int n, const T* __restrict__ L, const T* __restrict__ gL,
                               T* __restrict__ gA) {
    __shared__ T l[MAXN * (MAXN + 1)];
    __shared__ T lb[MAXN * (MAXN + 1)];   // adjoint of L, lower triangle
    __shared__ T ab[MAXN * (MAXN + 1)];   // adjoint of A, lower triangle
    const int k = threadIdx.x, ld = n + 1;
    for (int e = k; e < n * n; e += 64) {
        const int r = e / n, c = e % n;
        l[r * ld + c] = L[e];
        lb[r * ld + c] = c <= r ? gL[e] : T(0);
        ab[r * ld + c] = T(0);
    }
    __syncthreads();
    for (int j = n - 1; j >= 0; --j) {
        const T ljj = l[j * ld + j];
        for (int i = n - 1; i > j; --i) {
            // L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]) / L[j][j]
            const T t = lb[i * ld + j] / ljj;
            __syncthreads();
            if (k < j) {
                lb[i * ld + k] -= t * l[j * ld + k];
                lb[j * ld + k] -= t * l[i * ld + k];
            } else if (k == j) {
                ab[i * ld + j] = t;
                lb[j * ld + j] -= t * l[i * ld + j];
            }
            __syncthreads();
        }
        // L[j][j] = sqrt(A[j][j] - sum_{k<j} L[j][k]^2)
        const T t = lb[j * ld + j] / (T(2) * ljj);
        __syncthreads();
        if (k < j) lb[j * ld + k] -= T(2) * t * l[j * ld + k];
        else if (k == j) ab[j * ld + j] = t;
        __syncthreads();
    }
    for (int e = k; e < n * n; e += 64) {
        const int r = e / n, c = e % n;
        gA[e] = r == c ? ab[r * ld + r] : T(0.5) * (r > c ? ab[r * ld + c] : ab[c * ld + r]);
    }
}

// Q = Lq Lq^T + q_reg I,  R = Lr Lr^T + r_reg I  and  LR = chol(R)
// (kvae/state_space_model.py:112-146 and the factor the smoother's observation
// noise needs) in ONE launch of two single-wavefront workgroups: workgroup 0
// forms Q, workgroup 1 forms R and factors it.  As separate library calls these
// are seven launches of a few microseconds each per training step.  Forward only:
// callers that differentiate through the noise factors use the separate
// operations.
template <typename T, int MAXN>
__global__ void __launch_bounds__(64)
noise_factors_kernel(int n, int p, const T* __restrict__ Lq, const T* __restrict__ Lr, T q_reg,
                     T r_reg, T* __restrict__ Q, T* __restrict__ R, T* __restrict__ LR,
                     int* status) {
    __shared__ T f[MAXN * (MAXN + 1)];
    __shared__ T s[MAXN * (MAXN + 1)];
    const bool second = blockIdx.x == 1;
    const int d = second ? p : n, ld = d + 1, i = threadIdx.x;
    const T* F = second ? Lr : Lq;
    const T reg = second ? r_reg : q_reg;
    T* out = second ? R : Q;
    for (int e = i; e < d * d; e += 64) f[(e / d) * ld + e % d] = F[e];
    __syncthreads();
    for (int e = i; e < d * d; e += 64) {
        const int r = e / d, c = e % d;
        T acc = r == c ? reg : T(0);
        for (int k = 0; k < d; ++k) acc += f[r * ld + k] * f[c * ld + k];
        s[r * ld + c] = acc;
        out[e] = acc;
    }
    __syncthreads();
    if (!second || !LR) return;
    bool bad = false;
    for (int j = 0; j < d; ++j) {
        T acc = T(0);
        if (i >= j && i < d) {
            acc = s[i * ld + j];
            for (int k = 0; k < j; ++k) acc -= s[i * ld + k] * s[j * ld + k];
        }
        __syncthreads();
        if (i == j) {
            if (!(acc > T(0))) bad = true;
            s[j * ld + j] = sqrt(acc);
        }
        __syncthreads();
        if (i > j && i < d) s[i * ld + j] = acc / s[j * ld + j];
        __syncthreads();
    }
    for (int e = i; e < d * d; e += 64) {
        const int r = e / d, c = e % d;
        LR[e] = c <= r ? s[r * ld + c] : T(0);
    }
    if (bad && status) atomicOr(status, 8);
}

}  // namespace kvae

extern "C" {

int kvae_noise_factors(int n, int p, int dtype, const void* Lq, const void* Lr, double q_reg,
                       double r_reg, void* Q, void* R, void* LR, void* status, void* stream) {
    if (n < 1 || p < 1 || !Lq || !Lr || !Q || !R) return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KVAE_F32 && n <= 64 && p <= 64)
        hipLaunchKernelGGL((kvae::noise_factors_kernel<float, 64>), dim3(2), dim3(64), 0, s, n, p,
                           (const float*)Lq, (const float*)Lr, (float)q_reg, (float)r_reg, (float*)Q,
                           (float*)R, (float*)LR, (int*)status);
    else if (dtype == KVAE_F64 && n <= 32 && p <= 32)
        hipLaunchKernelGGL((kvae::noise_factors_kernel<double, 32>), dim3(2), dim3(64), 0, s, n, p,
                           (const double*)Lq, (const double*)Lr, q_reg, r_reg, (double*)Q, (double*)R,
                           (double*)LR, (int*)status);
    else
        return KVAE_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_cholesky_forward(int n, int dtype, const void* A, void* L, void* status, void* stream) {
    if (n < 1 || n > kvae::CHOL_MAX_N || !A || !L) return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KVAE_F32)
        hipLaunchKernelGGL((kvae::small_cholesky_forward_kernel<float, 64>), dim3(1), dim3(64), 0,
                           s, n, (const float*)A, (float*)L, (int*)status);
    else if (dtype == KVAE_F64 && n <= 32)
        hipLaunchKernelGGL((kvae::small_cholesky_forward_kernel<double, 32>), dim3(1), dim3(64), 0,
                           s, n, (const double*)A, (double*)L, (int*)status);
    else
        return KVAE_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_cholesky_backward(int n, int dtype, const void* L, const void* gL, void* gA,
                           void* stream) {
    if (n < 1 || n > kvae::CHOL_MAX_N || !L || !gL || !gA) return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == KVAE_F32)
        hipLaunchKernelGGL((kvae::small_cholesky_backward_kernel<float, 64>), dim3(1), dim3(64), 0,
                           s, n, (const float*)L, (const float*)gL, (float*)gA);
    else if (dtype == KVAE_F64 && n <= 32)
        hipLaunchKernelGGL((kvae::small_cholesky_backward_kernel<double, 32>), dim3(1), dim3(64), 0,
                           s, n, (const double*)L, (const double*)gL, (double*)gA);
    else
        return KVAE_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
