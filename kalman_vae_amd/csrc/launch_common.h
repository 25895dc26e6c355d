// Launch-side helpers shared by the lane-per-sequence and wavefront-per-sequence
// translation units: C-ABI struct -> kernel argument block conversion, the
// deterministic row-sum kernel for parameter gradients, small utilities.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "kalman_core.h"

namespace kvae {

constexpr int WAVE = 64;

// (n, p, K) combinations the small-state kernels (one or four lanes per sequence) are
// compiled for (an experiment build may define a shorter list on the command line)
#ifndef KVAE_DIM_LIST
#define KVAE_DIM_LIST(X) \
    X(4, 2, 3)           \
    X(2, 1, 2)           \
    X(3, 2, 2)           \
    X(4, 2, 8)           \
    X(8, 4, 4)
#endif

// Lane exchange inside a row of 16 lanes by a DPP control word (no LDS crossbar).
template <int CTRL>
__device__ __forceinline__ float row_exchange(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double row_exchange(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane_value(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ double lane_value(double x, int lane) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)b, lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes, the same value in every lane.  Four DPP butterfly stages make
// each row of 16 uniform (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror);
// the four row sums are then read with v_readlane.  The shuffle form of this (six
// ds_bpermute round trips per value) was 40 % of the ELBO backward kernel's cycles.
template <typename R>
__device__ __forceinline__ R wave_sum(R v) {
    v += row_exchange<0xB1>(v);
    v += row_exchange<0x4E>(v);
    v += row_exchange<0x141>(v);
    v += row_exchange<0x140>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

// out[c] = sum_r partial[r][c]; one workgroup per column, fixed summation
// order.  Columns are routed to up to six output tensors by offset.
struct ReduceSegments {
    void* out[6];
    int begin[7];  // begin[i]..begin[i+1] = columns of out[i]
    int count;
};

template <typename R>
__global__ void __launch_bounds__(256) reduce_rows_kernel(const R* partial, long rows, int len,
                                                         ReduceSegments seg) {
    __shared__ R sm[256 / WAVE];
    const int c = blockIdx.x;
    R s = R(0);
    for (long r = threadIdx.x; r < rows; r += 256) s += partial[r * len + c];
    s = wave_sum<R>(s);
    if ((threadIdx.x & (WAVE - 1)) == 0) sm[threadIdx.x / WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        R tot = R(0);
        for (int w = 0; w < 256 / WAVE; ++w) tot += sm[w];
        for (int i = 0; i < seg.count; ++i) {
            if (c >= seg.begin[i] && c < seg.begin[i + 1] && seg.out[i])
                static_cast<R*>(seg.out[i])[c - seg.begin[i]] = tot;
        }
    }
}

template <typename R>
FilterFwdArgs<R> make_args(const kvae_dims* d, const kvae_filter_fwd* io) {
    FilterFwdArgs<R> g;
    g.T = d->T;
    g.B = d->B;
    g.symmetrize = d->symmetrize;
    g.carry_per_sequence = io->carry_per_sequence;
    g.a = (const R*)io->a;
    g.alpha = (const R*)io->alpha;
    g.mask = (const R*)io->mask;
    g.A_K = (const R*)io->A_K;
    g.C_K = (const R*)io->C_K;
    g.Q = (const R*)io->Q;
    g.Rm = (const R*)io->R;
    g.m_in = (const R*)io->m_in;
    g.P_in = (const R*)io->P_in;
    g.m = (R*)io->mean_filt;
    g.Pf = (R*)io->cov_filt;
    g.mn = (R*)io->mean_pred;
    g.Pn = (R*)io->cov_pred;
    g.As = (R*)io->mat_As;
    g.Cs = (R*)io->mat_Cs;
    g.status = io->status;
    return g;
}

template <typename R>
SmoothFwdArgs<R> make_args(const kvae_dims* d, const kvae_smooth_fwd* io) {
    SmoothFwdArgs<R> g;
    g.T = d->T;
    g.B = d->B;
    g.symmetrize = d->symmetrize;
    g.m = (const R*)io->mean_filt;
    g.Pf = (const R*)io->cov_filt;
    g.mn = (const R*)io->mean_pred;
    g.Pn = (const R*)io->cov_pred;
    g.As = (const R*)io->mat_As;
    g.Cs = (const R*)io->mat_Cs;
    g.LR = (const R*)io->LR;
    g.eps_s = (const R*)io->eps_state;
    g.eps_o = (const R*)io->eps_obs;
    g.ms = (R*)io->mean_smooth;
    g.Ps = (R*)io->cov_smooth;
    g.zs = (R*)io->zs;
    g.ares = (R*)io->as_resampled;
    g.status = io->status;
    return g;
}

template <typename R>
ElboFwdArgs<R> make_args(const kvae_dims* d, const kvae_elbo_fwd* io) {
    ElboFwdArgs<R> g;
    g.T = d->T;
    g.B = d->B;
    g.a = (const R*)io->a;
    g.ms = (const R*)io->mean_smooth;
    g.Ps = (const R*)io->cov_smooth;
    g.As = (const R*)io->mat_As;
    g.Cs = (const R*)io->mat_Cs;
    g.Q = (const R*)io->Q;
    g.Rm = (const R*)io->R;
    g.m0 = (const R*)io->m0;
    g.P0 = (const R*)io->P0;
    g.eps = (const R*)io->eps;
    g.obs = (R*)io->obs;
    g.trans = (R*)io->trans;
    g.post = (R*)io->post;
    g.z = (R*)io->z;
    g.status = io->status;
    return g;
}

inline int launch_status() {
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

inline bool dims_ok(const kvae_dims* d) {
    return d && d->T >= 1 && d->B >= 1 && (d->dtype == KVAE_F32 || d->dtype == KVAE_F64);
}



template <typename R>
FilterBwdArgs<R> make_args(const kvae_dims* d, const kvae_filter_bwd* io, void* ws) {
    FilterBwdArgs<R> g;
    g.f = make_args<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_m = (const R*)io->g_mean_filt;
    g.g_P = (const R*)io->g_cov_filt;
    g.g_mn = (const R*)io->g_mean_pred;
    g.g_Pn = (const R*)io->g_cov_pred;
    g.g_As = (const R*)io->g_mat_As;
    g.g_Cs = (const R*)io->g_mat_Cs;
    g.d_a = (R*)io->d_a;
    g.d_alpha = (R*)io->d_alpha;
    g.d_m_in = (R*)io->d_m_in;
    g.d_P_in = (R*)io->d_P_in;
    g.partial = (R*)ws;
    g.bar_As = nullptr;
    g.bar_Cs = nullptr;
    return g;
}

template <typename R>
SmoothBwdArgs<R> make_args(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws) {
    SmoothBwdArgs<R> g;
    g.f = make_args<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_ms = (const R*)io->g_mean_smooth;
    g.g_Ps = (const R*)io->g_cov_smooth;
    g.g_zs = (const R*)io->g_zs;
    g.g_ares = (const R*)io->g_as_resampled;
    g.d_m = (R*)io->d_mean_filt;
    g.d_P = (R*)io->d_cov_filt;
    g.d_mn = (R*)io->d_mean_pred;
    g.d_Pn = (R*)io->d_cov_pred;
    g.d_As = (R*)io->d_mat_As;
    g.d_Cs = (R*)io->d_mat_Cs;
    g.partial = io->d_LR ? (R*)ws : nullptr;
    return g;
}

template <typename R>
ElboBwdArgs<R> make_args(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws) {
    ElboBwdArgs<R> g;
    g.f = make_args<R>(d, &io->fwd);
    g.g_obs = (const R*)io->g_obs;
    g.g_trans = (const R*)io->g_trans;
    g.g_post = (const R*)io->g_post;
    g.g_scale = (const R*)io->g_scale;
    g.g_scale_stride = io->g_scale_stride;
    g.w_obs = (R)io->w_obs;
    g.w_trans = (R)io->w_trans;
    g.w_post = (R)io->w_post;
    g.d_a = (R*)io->d_a;
    g.d_ms = (R*)io->d_mean_smooth;
    g.d_Ps = (R*)io->d_cov_smooth;
    g.d_As = (R*)io->d_mat_As;
    g.d_Cs = (R*)io->d_mat_Cs;
    g.partial = (R*)ws;
    return g;
}

// Launch reduce_rows_kernel: columns [begin[i], begin[i+1]) of the (rows, len)
// partial-sum matrix go to outs[i] (null = skip).
template <typename R>
void launch_reduce(const void* ws, long rows, int len, int count, void* const* outs,
                   const int* begin, hipStream_t s) {
    ReduceSegments seg;
    seg.count = count;
    for (int i = 0; i < 6; ++i) seg.out[i] = i < count ? outs[i] : nullptr;
    for (int i = 0; i < 7; ++i) seg.begin[i] = i <= count ? begin[i] : len;
    hipLaunchKernelGGL((reduce_rows_kernel<R>), dim3(len), dim3(256), 0, s, (const R*)ws, rows, len,
                       seg);
}

template <typename R, int N, int P, int K>
void reduce_filter_grads(const kvae_filter_bwd* io, const void* ws, long rows, hipStream_t s) {
    using PG = FilterParamGrads<N, P, K>;
    void* outs[4] = {io->d_A_K, io->d_C_K, io->d_Q, io->d_R};
    const int begin[5] = {PG::OFF_A, PG::OFF_C, PG::OFF_Q, PG::OFF_R, PG::LEN};
    launch_reduce<R>(ws, rows, PG::LEN, 4, outs, begin, s);
}

template <typename R, int N, int P>
void reduce_elbo_grads(const kvae_elbo_bwd* io, const void* ws, long rows, hipStream_t s) {
    using PG = ElboParamGrads<N, P>;
    void* outs[4] = {io->d_Q, io->d_R, io->d_m0, io->d_P0};
    const int begin[5] = {PG::OFF_Q, PG::OFF_R, PG::OFF_M0, PG::OFF_P0, PG::LEN};
    launch_reduce<R>(ws, rows, PG::LEN, 4, outs, begin, s);
}

template <typename R, int P>
void reduce_smooth_grads(const kvae_smooth_bwd* io, const void* ws, long rows, hipStream_t s) {
    void* outs[1] = {io->d_LR};
    const int begin[2] = {0, P * P};
    launch_reduce<R>(ws, rows, P * P, 1, outs, begin, s);
}

}  // namespace kvae
