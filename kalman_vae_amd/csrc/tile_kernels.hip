// Register-tile kernels for large state dimensions (fp32, z_dim and a_dim
// multiples of 8 up to 32): one wavefront per sequence, all matrices in VGPRs,
// products on v_mfma_f32_32x32x2_f32 (kalman_tile.h).  The wavefront-per-sequence
// translation unit (wave_kernels.hip) tries these first and falls back to its
// LDS-resident kernels for other shapes and for fp64; KVAE_LARGE_STATE=lds in
// the environment forces the LDS kernels (used by the tests to compare the two).
#include <cstdint>
#include <cstdlib>
#include <initializer_list>
#include <cstring>

#include "kalman_tile.h"
#include "device_info.h"
#include "launch_common.h"
#include "wave_api.h"

namespace kvae {

constexpr int TILE_LDS_FLOATS = 32 * 33;

template <int N, int P>
__global__ void __launch_bounds__(64) tile_elbo_forward_kernel(ElboFwdArgs<float> g) {
    __shared__ float lds[TILE_LDS_FLOATS];
    // blockIdx.y: which piece of the sequence (gridDim.y pieces of equal length, the last shorter)
    const int len = (g.T + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t0 = (int)blockIdx.y * len, t1 = t0 + len < g.T ? t0 + len : g.T;
    if (t0 < t1) tile::elbo_forward_sequence<N, P>(g, blockIdx.x, lds, t0, t1);
}

// Two wavefronts per SIMD (256 registers each, 8 workgroups' LDS per compute unit): the three
// parameter-gradient accumulators are tiles in LDS for that.
template <int N, int P>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
tile_elbo_backward_kernel(ElboBwdArgs<float> g) {
    __shared__ float lds[TILE_LDS_FLOATS + 3 * 1024];
    const int len = (g.f.T + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t0 = (int)blockIdx.y * len, t1 = t0 + len < g.f.T ? t0 + len : g.f.T;
    // (a piece without steps still writes its row of zeros: the row sum reads every row)
    tile::elbo_backward_sequence<N, P>(g, blockIdx.x, lds, lds + TILE_LDS_FLOATS, t0 < t1 ? t0 : t1,
                                       t1, (int)blockIdx.y);
}

template <int N, int P>
__global__ void __launch_bounds__(64) tile_smooth_forward_kernel(SmoothFwdArgs<float> g) {
    __shared__ float lds[TILE_LDS_FLOATS];
    tile::smooth_forward_sequence<N, P>(g, blockIdx.x, lds);
}

template <int N, int P>
__global__ void __launch_bounds__(64) tile_smooth_backward_kernel(SmoothBwdArgs<float> g) {
    __shared__ float lds[TILE_LDS_FLOATS];
    tile::smooth_backward_sequence<N, P>(g, blockIdx.x, lds);
}

template <int N, int P>
__global__ void __launch_bounds__(64) tile_filter_forward_kernel(FilterFwdArgs<float> g, int K) {
    __shared__ float lds[TILE_LDS_FLOATS];
    tile::filter_forward_sequence<N, P>(g, K, blockIdx.x, lds);
}

template <int N, int P>
__global__ void __launch_bounds__(64) tile_filter_backward_kernel(FilterBwdArgs<float> g) {
    __shared__ float lds[TILE_LDS_FLOATS];
    tile::filter_backward_sequence<N, P>(g, blockIdx.x, lds);
}

// (n, p) compiled for the register-tile path
#define KVAE_TILE_DIM_LIST(X) \
    X(32, 16)                 \
    X(16, 8)

// the tile kernels read and write matrix rows 16 bytes at a time
static bool aligned16(std::initializer_list<const void*> ptrs) {
    for (const void* p : ptrs)
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    return true;
}

bool tile_enabled() {
    const char* e = std::getenv("KVAE_LARGE_STATE");
    return !(e && std::strcmp(e, "lds") == 0);
}

int tile_supported(int n, int p, int dtype) {
    if (dtype != KVAE_F32 || !tile_enabled()) return 0;
#define KVAE_SUP(n_, p_) \
    if (n == n_ && p == p_) return 1;
    KVAE_TILE_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return 0;
}

int tile_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s) {
    if (!tile_supported(d->n, d->p, d->dtype)) return KVAE_E_UNSUPPORTED;
    FilterFwdArgs<float> g = make_args<float>(d, io);
    if (!aligned16({g.A_K, g.C_K, g.As, g.Cs})) return KVAE_E_UNSUPPORTED;
#define KVAE_CASE(n_, p_)                                                                        \
    if (d->n == n_ && d->p == p_) {                                                              \
        hipLaunchKernelGGL((tile_filter_forward_kernel<n_, p_>), dim3(d->B), dim3(64), 0, s, g,   \
                           d->K);                                                                \
        return launch_status();                                                                  \
    }
    KVAE_TILE_DIM_LIST(KVAE_CASE)
#undef KVAE_CASE
    return KVAE_E_UNSUPPORTED;
}

// the per-sequence recursion of the filter backward only; the caller runs the
// reductions and the mixing-adjoint passes over bar_As / bar_Cs
int tile_filter_backward_main(const kvae_dims* d, const FilterBwdArgs<float>& g, hipStream_t s) {
    if (!tile_supported(d->n, d->p, d->dtype) || !g.f.As || !g.f.Cs ||
        !aligned16({g.f.As, g.f.Cs}))
        return KVAE_E_UNSUPPORTED;
#define KVAE_CASE(n_, p_)                                                                        \
    if (d->n == n_ && d->p == p_) {                                                              \
        hipLaunchKernelGGL((tile_filter_backward_kernel<n_, p_>), dim3(d->B), dim3(64), 0, s, g); \
        return KVAE_OK;                                                                          \
    }
    KVAE_TILE_DIM_LIST(KVAE_CASE)
#undef KVAE_CASE
    return KVAE_E_UNSUPPORTED;
}

int tile_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s) {
    if (!tile_supported(d->n, d->p, d->dtype)) return KVAE_E_UNSUPPORTED;
    SmoothFwdArgs<float> g = make_args<float>(d, io);
    if (!aligned16({g.As, g.Cs, g.LR})) return KVAE_E_UNSUPPORTED;
#define KVAE_CASE(n_, p_)                                                                        \
    if (d->n == n_ && d->p == p_) {                                                              \
        hipLaunchKernelGGL((tile_smooth_forward_kernel<n_, p_>), dim3(d->B), dim3(64), 0, s, g);  \
        return launch_status();                                                                  \
    }
    KVAE_TILE_DIM_LIST(KVAE_CASE)
#undef KVAE_CASE
    return KVAE_E_UNSUPPORTED;
}

int tile_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws, size_t ws_bytes,
                         hipStream_t s) {
    if (!tile_supported(d->n, d->p, d->dtype)) return KVAE_E_UNSUPPORTED;
#define KVAE_CASE(n_, p_)                                                                         \
    if (d->n == n_ && d->p == p_) {                                                               \
        if (io->d_LR && ws_bytes < (size_t)d->B * p_ * p_ * sizeof(float)) return KVAE_E_WORKSPACE; \
        SmoothBwdArgs<float> g = make_args<float>(d, io, ws);                                     \
        if (!aligned16({g.f.As, g.f.Cs, g.f.LR})) return KVAE_E_UNSUPPORTED;                      \
        hipLaunchKernelGGL((tile_smooth_backward_kernel<n_, p_>), dim3(d->B), dim3(64), 0, s, g);  \
        if (io->d_LR) reduce_smooth_grads<float, p_>(io, ws, d->B, s);                            \
        return launch_status();                                                                   \
    }
    KVAE_TILE_DIM_LIST(KVAE_CASE)
#undef KVAE_CASE
    return KVAE_E_UNSUPPORTED;
}

// Pieces a sequence's ELBO terms are cut into: enough wavefronts for three per SIMD (the forward
// kernel's register count allows three), pieces of at least 32 steps (each piece factors one
// covariance more than it owns).
static unsigned elbo_pieces(const kvae_dims* d, int waves_per_simd = 3) {
    const long simds = 4L * device_compute_units();
    long pieces = (waves_per_simd * simds + d->B - 1) / d->B;
    if (pieces > d->T / 32) pieces = d->T / 32;
    if (pieces > 8) pieces = 8;
    return (unsigned)(pieces < 1 ? 1 : pieces);
}

// the backward kernel runs two wavefronts per SIMD; one row of parameter-gradient sums per
// sequence and piece
int tile_elbo_backward_pieces(const kvae_dims* d) {
    return tile_supported(d->n, d->p, d->dtype) ? (int)elbo_pieces(d, 2) : 1;
}

int tile_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, hipStream_t s) {
    if (!tile_supported(d->n, d->p, d->dtype)) return KVAE_E_UNSUPPORTED;
    ElboFwdArgs<float> g = make_args<float>(d, io);
    if (!aligned16({g.As, g.Cs})) return KVAE_E_UNSUPPORTED;
#define KVAE_CASE(n_, p_)                                                                       \
    if (d->n == n_ && d->p == p_) {                                                             \
        hipLaunchKernelGGL((tile_elbo_forward_kernel<n_, p_>), dim3(d->B, elbo_pieces(d)),      \
                           dim3(64), 0, s, g);                                                  \
        return launch_status();                                                                 \
    }
    KVAE_TILE_DIM_LIST(KVAE_CASE)
#undef KVAE_CASE
    return KVAE_E_UNSUPPORTED;
}

int tile_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    if (!tile_supported(d->n, d->p, d->dtype)) return KVAE_E_UNSUPPORTED;
#define KVAE_CASE(n_, p_)                                                                        \
    if (d->n == n_ && d->p == p_) {                                                              \
        using PG = ElboParamGrads<n_, p_>;                                                       \
        const unsigned pieces = elbo_pieces(d, 2);                                               \
        if (ws_bytes < (size_t)d->B * pieces * PG::LEN * sizeof(float)) return KVAE_E_WORKSPACE; \
        ElboBwdArgs<float> g = make_args<float>(d, io, ws);                                      \
        if (!aligned16({g.f.As, g.f.Cs})) return KVAE_E_UNSUPPORTED;                             \
        hipLaunchKernelGGL((tile_elbo_backward_kernel<n_, p_>), dim3(d->B, pieces), dim3(64), 0, \
                           s, g);                                                                \
        reduce_elbo_grads<float, n_, p_>(io, ws, (long)d->B * pieces, s);                        \
        return launch_status();                                                                  \
    }
    KVAE_TILE_DIM_LIST(KVAE_CASE)
#undef KVAE_CASE
    return KVAE_E_UNSUPPORTED;
}

}  // namespace kvae
