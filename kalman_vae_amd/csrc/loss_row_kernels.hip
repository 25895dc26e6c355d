// gfx950 kernel of the fused Kalman objective in its row form (kalman_row.h): one sequence
// over the 16 lanes of a DPP row, four sequences per wavefront, one wavefront per workgroup.
// 4096 sequences are 1024 wavefronts: one per SIMD of the chip.  The per-step state that the
// smoother and the adjoint sweeps read again stays in LDS; global memory sees a, alpha, eps
// on the way in and the term sums, d_a, d_alpha and one row of parameter-gradient sums per
// wavefront on the way out.
//
// Called from loss_kernels.hip (kvae_kalman_loss_forward / _backward / _value_and_grad).
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "kalman_row.h"
#include "launch_common.h"
#include "device_info.h"
#include "wave_api.h"

namespace kvae {

namespace {

constexpr int ROW_SEQS = 4;           // sequences per wavefront
constexpr int ROW_LDS_LIMIT = 65536;  // bytes of LDS one workgroup may use here

// this lane's share of the memory accesses of kalman_row.h
template <bool GRAD>
struct DevRowEnv {
    using SL = row::Slots<GRAD>;
    using Loc = float*;
    float* lds;
    int T, l16, li, lj, sq;
    bool active;

    __device__ __forceinline__ float scalar(const float* p) const { return *p; }
    template <int LEN>
    __device__ __forceinline__ float vec_c(const float* p) const {
        const float v = p[lj < LEN ? lj : 0];
        return lj < LEN ? v : 0.f;
    }
    template <int LEN>
    __device__ __forceinline__ float vec_r(const float* p) const {
        const float v = p[li < LEN ? li : 0];
        return li < LEN ? v : 0.f;
    }
    template <int ROWS, int COLS, int LAY, bool PAD>
    __device__ __forceinline__ float mat(const float* p) const {
        const int r = LAY == row::L_NAT ? li : LAY == row::L_TRN ? lj : LAY == row::L_SKEW ? li : ((li + lj) & 3);
        const int c = LAY == row::L_NAT ? lj : LAY == row::L_TRN ? li : LAY == row::L_SKEW ? ((li + lj) & 3) : li;
        const bool in = r < ROWS && c < COLS;
        const float v = p[in ? r * COLS + c : 0];
        return in ? v : ((PAD && r == c) ? 1.f : 0.f);
    }
    __device__ __forceinline__ void put_scalar(float* p, float x) const {
        if (active && l16 == 0) *p = x;
    }
    // a step is STEP floats per sequence: the wavefront's four sequences side by side per slot
    __device__ __forceinline__ Loc at_mat(int t, int slot) const {
        return lds + (long)t * (SL::STEP * ROW_SEQS) + slot * ROW_SEQS + sq * 16;
    }
    __device__ __forceinline__ Loc at_vec(int t, int slot) const {
        return lds + (long)t * (SL::STEP * ROW_SEQS) + slot * ROW_SEQS + sq * 4;
    }
    __device__ __forceinline__ Loc scr(int i) const {
        return lds + (long)T * (SL::STEP * ROW_SEQS) + i * (16 * ROW_SEQS) + sq * 16;
    }
    __device__ __forceinline__ void st_mat(Loc q, float x) const { q[l16] = x; }
    __device__ __forceinline__ void st_mat_t(Loc q, float x) const { q[4 * lj + li] = x; }
    __device__ __forceinline__ float ld_mat(Loc q) const { return q[l16]; }
    __device__ __forceinline__ float ld_mat_t(Loc q) const { return q[4 * lj + li]; }
    __device__ __forceinline__ row::Q4<float> ld_rows4(Loc q) const {
        const float4 v = *reinterpret_cast<const float4*>(q + 4 * lj);
        row::Q4<float> y;
        y.v[0] = v.x; y.v[1] = v.y; y.v[2] = v.z; y.v[3] = v.w;
        return y;
    }
    __device__ __forceinline__ row::Q4<float> ld_cols4(Loc q) const {
        row::Q4<float> y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y.v[k] = q[4 * k + lj];
        return y;
    }
    template <int KK>
    __device__ __forceinline__ float ld_colc(Loc q) const { return q[4 * lj + KK]; }
    __device__ __forceinline__ void st_vec(Loc q, float xc) const {
        if (li == 0) q[lj] = xc;
    }
    __device__ __forceinline__ float ld_vec_c(Loc q) const { return q[lj]; }
    __device__ __forceinline__ float ld_vec_r(Loc q) const { return q[li]; }
    // orders this wavefront's LDS accesses: a slot written by one lane is read by another
    __device__ __forceinline__ void fence() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// value of x in the lane 16 * q + l16 for q = 0..3, summed
__device__ __forceinline__ float sum_over_sequences(float x) {
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// sums_partial: (rows, 3) or null; grad_partial: (rows, LossParamGrads::LEN), GRAD only
template <int N, int P, int K, bool GRAD, bool SYM>
__global__ void __launch_bounds__(64) loss_row_kernel(LossBwdArgs<float> g, float* sums_partial,
                                                      float* grad_partial) {
    extern __shared__ __attribute__((aligned(16))) float row_lds[];
    using PG = LossParamGrads<N, P, K>;
    const int lane = threadIdx.x;
    DevRowEnv<GRAD> env;
    env.lds = row_lds;
    env.T = g.f.T;
    env.l16 = lane & 15;
    env.li = env.l16 >> 2;
    env.lj = env.l16 & 3;
    env.sq = lane >> 4;
    long b = (long)blockIdx.x * ROW_SEQS + env.sq;
    env.active = b < g.f.B;
    if (!env.active) b = g.f.B - 1;   // stays in step with the wavefront, writes nothing

    float sums[3] = {0.f, 0.f, 0.f};
    row::RowAcc<float, P, K> acc;
    row::zero_acc(acc);
    int st = 0;
    row::loss_row_sequence<float, float, N, P, K, GRAD, SYM>(env, g, b, sums, acc, st);
    if (st && env.active && g.f.status) atomicOr(g.f.status, st);

    if (sums_partial) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float v = sum_over_sequences(env.active ? sums[e] : 0.f);
            if (lane == 0) sums_partial[(long)blockIdx.x * 3 + e] = v;
        }
    }
    if constexpr (GRAD) {
        float* out = grad_partial + (long)blockIdx.x * PG::LEN;
        const float on = env.active ? 1.f : 0.f;
        const bool first = lane < 16;
        const int i = env.li, j = env.lj;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float a = sum_over_sequences(on * (acc.AK[k] + row::tr(acc.AKT[k])));
            if (first && i < N && j < N) out[PG::OFF_A + k * N * N + i * N + j] = a;
#pragma unroll
            for (int r = 0; r < P; ++r) {
                const float c = sum_over_sequences(on * acc.CKc[k][r]);
                if (first && i == 0 && j < N) out[PG::OFF_C + (k * P + r) * N + j] = c;
            }
        }
        const float q = sum_over_sequences(on * acc.Q);
        if (first && i < N && j < N) out[PG::OFF_Q + i * N + j] = q;
#pragma unroll
        for (int r = 0; r < P; ++r) {
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const float v = sum_over_sequences(on * acc.Rs[r][c]);
                if (lane == 0) out[PG::OFF_R + r * P + c] = v;
            }
        }
        const float m0 = sum_over_sequences(on * acc.m0c);
        if (first && i == 0 && j < N) out[PG::OFF_M0 + j] = m0;
        const float p0 = sum_over_sequences(on * acc.P0);
        if (first && i < N && j < N) out[PG::OFF_P0 + i * N + j] = p0;
    }
}

// The sums over the wavefronts' rows, in a fixed order, as ONE launch after the main kernel:
// workgroup c < len adds column c of the (rows, len) parameter-gradient partials and routes it
// to its output tensor; workgroup len (present when sums != null) adds the three term columns
// and writes out (4) = [ w0 * sum obs, w1 * sum trans, w2 * sum post, their total ].
__global__ void __launch_bounds__(256) loss_row_sums_kernel(const float* grads, const float* sums,
                                                           long rows, int len, ReduceSegments seg,
                                                           float w0, float w1, float w2,
                                                           float* out) {
    __shared__ float sm[3][256 / WAVE];
    const int c = blockIdx.x;
    const int wave = threadIdx.x / WAVE;
    const bool lead = (threadIdx.x & (WAVE - 1)) == 0;
    if (c < len) {
        float s = 0.f;
        for (long r = threadIdx.x; r < rows; r += 256) s += grads[r * len + c];
        s = wave_sum<float>(s);
        if (lead) sm[0][wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int w = 0; w < 256 / WAVE; ++w) tot += sm[0][w];
            for (int i = 0; i < seg.count; ++i) {
                if (c >= seg.begin[i] && c < seg.begin[i + 1] && seg.out[i])
                    static_cast<float*>(seg.out[i])[c - seg.begin[i]] = tot;
            }
        }
        return;
    }
    float s[3] = {0.f, 0.f, 0.f};
    for (long r = threadIdx.x; r < rows; r += 256) {
#pragma unroll
        for (int e = 0; e < 3; ++e) s[e] += sums[r * 3 + e];
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float v = wave_sum<float>(s[e]);
        if (lead) sm[e][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            tot[e] = 0.f;
            for (int w = 0; w < 256 / WAVE; ++w) tot[e] += sm[e][w];
        }
        out[0] = w0 * tot[0];
        out[1] = w1 * tot[1];
        out[2] = w2 * tot[2];
        out[3] = (out[0] + out[1]) + out[2];
    }
}

template <bool GRAD>
size_t row_lds_bytes(int T) {
    return ((size_t)T * row::Slots<GRAD>::STEP + row::SCRATCH * 16) * ROW_SEQS * sizeof(float);
}

LossBwdArgs<float> make_row_args(const kvae_dims* d, const kvae_loss_fwd* io) {
    LossBwdArgs<float> g = {};
    g.f.T = d->T;
    g.f.B = d->B;
    g.f.symmetrize = d->symmetrize;
    g.f.a = (const float*)io->a;
    g.f.alpha = (const float*)io->alpha;
    g.f.mask = (const float*)io->mask;
    g.f.A_K = (const float*)io->A_K;
    g.f.C_K = (const float*)io->C_K;
    g.f.Q = (const float*)io->Q;
    g.f.Rm = (const float*)io->R;
    g.f.m0 = (const float*)io->m0;
    g.f.P0 = (const float*)io->P0;
    g.f.eps = (const float*)io->eps;
    g.f.terms = (float*)io->terms;
    g.f.status = io->status;
    g.burn_in = d->burn_in;
    g.w_obs = (float)io->w_obs;
    g.w_trans = (float)io->w_trans;
    g.w_post = (float)io->w_post;
    return g;
}

template <int N, int P, int K>
int row_forward(const kvae_dims* d, const kvae_loss_fwd* io, void* ws, size_t ws_bytes,
                hipStream_t s) {
    const long rows = row_loss_rows(d->B);
    if (ws_bytes < (size_t)rows * 3 * sizeof(float)) return KVAE_E_WORKSPACE;
    LossBwdArgs<float> g = make_row_args(d, io);
    if (d->symmetrize)
        hipLaunchKernelGGL((loss_row_kernel<N, P, K, false, true>), dim3(rows), dim3(64),
                           row_lds_bytes<false>(d->T), s, g, (float*)ws, (float*)nullptr);
    else
        hipLaunchKernelGGL((loss_row_kernel<N, P, K, false, false>), dim3(rows), dim3(64),
                           row_lds_bytes<false>(d->T), s, g, (float*)ws, (float*)nullptr);
    ReduceSegments none = {};
    hipLaunchKernelGGL(loss_row_sums_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr,
                       (const float*)ws, rows, 0, none, (float)io->w_obs, (float)io->w_trans,
                       (float)io->w_post, (float*)io->out);
    return launch_status();
}

template <int N, int P, int K>
int row_backward(const kvae_dims* d, const kvae_loss_bwd* io, void* ws, size_t ws_bytes,
                 hipStream_t s, bool with_value) {
    using PG = LossParamGrads<N, P, K>;
    const long rows = row_loss_rows(d->B);
    if (ws_bytes < (size_t)rows * (PG::LEN + 3) * sizeof(float)) return KVAE_E_WORKSPACE;
    kvae_loss_fwd fio = io->fwd;
    if (!with_value) {
        fio.terms = nullptr;
        fio.status = nullptr;
    }
    LossBwdArgs<float> g = make_row_args(d, &fio);
    g.g_scale = (const float*)io->g_scale;
    g.g_scale_stride = io->g_scale_stride;
    g.d_a = (float*)io->d_a;
    g.d_alpha = (float*)io->d_alpha;
    float* grad_partial = (float*)ws;
    float* sums_partial = with_value ? grad_partial + rows * PG::LEN : nullptr;
    if (d->symmetrize)
        hipLaunchKernelGGL((loss_row_kernel<N, P, K, true, true>), dim3(rows), dim3(64),
                           row_lds_bytes<true>(d->T), s, g, sums_partial, grad_partial);
    else
        hipLaunchKernelGGL((loss_row_kernel<N, P, K, true, false>), dim3(rows), dim3(64),
                           row_lds_bytes<true>(d->T), s, g, sums_partial, grad_partial);
    void* outs[6] = {io->d_A_K, io->d_C_K, io->d_Q, io->d_R, io->d_m0, io->d_P0};
    const int begin[7] = {PG::OFF_A, PG::OFF_C, PG::OFF_Q, PG::OFF_R, PG::OFF_M0, PG::OFF_P0, PG::LEN};
    ReduceSegments seg;
    seg.count = 6;
    for (int i = 0; i < 6; ++i) seg.out[i] = outs[i];
    for (int i = 0; i < 7; ++i) seg.begin[i] = begin[i];
    hipLaunchKernelGGL(loss_row_sums_kernel, dim3(PG::LEN + (with_value ? 1 : 0)), dim3(256), 0, s,
                       (const float*)grad_partial, (const float*)sums_partial, rows, (int)PG::LEN,
                       seg, (float)fio.w_obs, (float)fio.w_trans, (float)fio.w_post,
                       (float*)fio.out);
    return launch_status();
}

}  // namespace

#ifndef KVAE_ROW_DIM_LIST   // (an experiment build may define a shorter list)
#define KVAE_ROW_DIM_LIST(X) X(4, 2, 3) X(2, 1, 2) X(3, 2, 2) X(4, 2, 8)
#endif

long row_loss_rows(long B) { return (B + ROW_SEQS - 1) / ROW_SEQS; }

// Sequences whose workgroups are all resident at once on the current device.  A compute unit
// holds as many workgroups as its 160 KB of LDS have room for (2 KB granules), and two
// wavefronts per SIMD are as far as this form was measured to gain.
long row_loss_resident_sequences(int T) {
    const size_t lds = (row_lds_bytes<true>(T) + 2047) / 2048 * 2048;
    long per_cu = (long)(160 * 1024 / lds);
    if (per_cu > 8) per_cu = 8;
    return (long)ROW_SEQS * device_compute_units() * per_cu;
}

int row_loss_supported(int n, int p, int K, int dtype, int T, int grad) {
    if (dtype != KVAE_F32) return 0;
    if ((grad ? row_lds_bytes<true>(T) : row_lds_bytes<false>(T)) > (size_t)ROW_LDS_LIMIT) return 0;
#define KVAE_SUP(n_, p_, K_) \
    if (n == n_ && p == p_ && K == K_) return 1;
    KVAE_ROW_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return 0;
}

size_t row_loss_workspace_bytes(const kvae_dims* d) {
    const long n = d->n, p = d->p, K = d->K;
    return (size_t)row_loss_rows(d->B) * (K * n * n + K * p * n + n * n + p * p + n + n * n + 3) *
           sizeof(float);
}

int row_loss_forward(const kvae_dims* d, const kvae_loss_fwd* io, void* ws, size_t ws_bytes,
                     hipStream_t s) {
#define KVAE_ROW_CASE(n_, p_, K_) \
    if (d->n == n_ && d->p == p_ && d->K == K_) return row_forward<n_, p_, K_>(d, io, ws, ws_bytes, s);
    KVAE_ROW_DIM_LIST(KVAE_ROW_CASE)
#undef KVAE_ROW_CASE
    return KVAE_E_UNSUPPORTED;
}

int row_loss_backward(const kvae_dims* d, const kvae_loss_bwd* io, void* ws, size_t ws_bytes,
                      hipStream_t s, int with_value) {
#define KVAE_ROW_CASE(n_, p_, K_)                          \
    if (d->n == n_ && d->p == p_ && d->K == K_)            \
        return row_backward<n_, p_, K_>(d, io, ws, ws_bytes, s, with_value != 0);
    KVAE_ROW_DIM_LIST(KVAE_ROW_CASE)
#undef KVAE_ROW_CASE
    return KVAE_E_UNSUPPORTED;
}

}  // namespace kvae
