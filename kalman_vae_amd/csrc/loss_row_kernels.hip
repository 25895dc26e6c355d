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
#include "wave_api.h"

namespace kvae {

namespace {

constexpr int ROW_SEQS = 4;           // sequences per wavefront
constexpr int ROW_LDS_LIMIT = 65536;  // bytes of LDS one workgroup may use here

// this lane's share of the memory accesses of kalman_row.h
template <bool GRAD>
struct DevRowEnv {
    using SL = row::Slots<GRAD>;
    float* lds;
    int l16, li, lj, sq;
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
    template <int LEN>
    __device__ __forceinline__ void put_vec_c(float* p, float xc) const {
        if (active && li == 0 && lj < LEN) p[lj] = xc;
    }
    __device__ __forceinline__ float* step(int t) const { return lds + (long)t * (SL::STEP * ROW_SEQS); }
    __device__ __forceinline__ void st_mat(int t, int slot, float x) const {
        step(t)[slot * ROW_SEQS + sq * 16 + l16] = x;
    }
    template <int LAY>
    __device__ __forceinline__ float ld_mat(int t, int slot) const {
        const int off = LAY == row::L_NAT ? l16 : 4 * lj + li;
        static_assert(LAY == row::L_NAT || LAY == row::L_TRN, "stored matrices are read plain or transposed");
        return step(t)[slot * ROW_SEQS + sq * 16 + off];
    }
    __device__ __forceinline__ void st_vec(int t, int slot, float xc) const {
        if (li == 0) step(t)[slot * ROW_SEQS + sq * 4 + lj] = xc;
    }
    __device__ __forceinline__ float ld_vec_c(int t, int slot) const {
        return step(t)[slot * ROW_SEQS + sq * 4 + lj];
    }
    __device__ __forceinline__ float ld_vec_r(int t, int slot) const {
        return step(t)[slot * ROW_SEQS + sq * 4 + li];
    }
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
template <int N, int P, int K, bool GRAD>
__global__ void __launch_bounds__(64) loss_row_kernel(LossBwdArgs<float> g, float* sums_partial,
                                                      float* grad_partial) {
    extern __shared__ float row_lds[];
    using PG = LossParamGrads<N, P, K>;
    const int lane = threadIdx.x;
    DevRowEnv<GRAD> env;
    env.lds = row_lds;
    env.l16 = lane & 15;
    env.li = env.l16 >> 2;
    env.lj = env.l16 & 3;
    env.sq = lane >> 4;
    long b = (long)blockIdx.x * ROW_SEQS + env.sq;
    env.active = b < g.f.B;
    if (!env.active) b = g.f.B - 1;   // stays in step with the wavefront, writes nothing

    float sums[3] = {0.f, 0.f, 0.f};
    row::RowAcc<float, K> acc;
    row::zero_acc(acc);
    int st = 0;
    row::loss_row_sequence<float, float, N, P, K, GRAD>(env, g, b, sums, acc, st);
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
            const float c = sum_over_sequences(on * acc.CK[k]);
            if (first && i < P && j < N) out[PG::OFF_C + k * P * N + i * N + j] = c;
        }
        const float q = sum_over_sequences(on * acc.Q);
        if (first && i < N && j < N) out[PG::OFF_Q + i * N + j] = q;
        const float r = sum_over_sequences(on * acc.Rm);
        if (first && i < P && j < P) out[PG::OFF_R + i * P + j] = r;
        const float m0 = sum_over_sequences(on * acc.m0c);
        if (first && i == 0 && j < N) out[PG::OFF_M0 + j] = m0;
        const float p0 = sum_over_sequences(on * acc.P0);
        if (first && i < N && j < N) out[PG::OFF_P0 + i * N + j] = p0;
    }
}

// out (4) = [ w0 * sum_r partial[r][0], w1 * ..., w2 * ..., their total ], fixed order.
__global__ void __launch_bounds__(256) loss_row_finish_kernel(const float* partial, long rows,
                                                             float w0, float w1, float w2,
                                                             float* out) {
    __shared__ float sm[3][256 / WAVE];
    float s[3] = {0.f, 0.f, 0.f};
    for (long r = threadIdx.x; r < rows; r += 256) {
#pragma unroll
        for (int e = 0; e < 3; ++e) s[e] += partial[r * 3 + e];
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float v = wave_sum<float>(s[e]);
        if ((threadIdx.x & (WAVE - 1)) == 0) sm[e][threadIdx.x / WAVE] = v;
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
    return (size_t)T * row::Slots<GRAD>::STEP * ROW_SEQS * sizeof(float);
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
    hipLaunchKernelGGL((loss_row_kernel<N, P, K, false>), dim3(rows), dim3(64),
                       row_lds_bytes<false>(d->T), s, g, (float*)ws, (float*)nullptr);
    hipLaunchKernelGGL(loss_row_finish_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, rows,
                       (float)io->w_obs, (float)io->w_trans, (float)io->w_post, (float*)io->out);
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
    hipLaunchKernelGGL((loss_row_kernel<N, P, K, true>), dim3(rows), dim3(64),
                       row_lds_bytes<true>(d->T), s, g, sums_partial, grad_partial);
    if (with_value)
        hipLaunchKernelGGL(loss_row_finish_kernel, dim3(1), dim3(256), 0, s,
                           (const float*)sums_partial, rows, (float)fio.w_obs, (float)fio.w_trans,
                           (float)fio.w_post, (float*)fio.out);
    void* outs[6] = {io->d_A_K, io->d_C_K, io->d_Q, io->d_R, io->d_m0, io->d_P0};
    const int begin[7] = {PG::OFF_A, PG::OFF_C, PG::OFF_Q, PG::OFF_R, PG::OFF_M0, PG::OFF_P0, PG::LEN};
    launch_reduce<float>(grad_partial, rows, PG::LEN, 6, outs, begin, s);
    return launch_status();
}

}  // namespace

#define KVAE_ROW_DIM_LIST(X) X(4, 2, 3) X(2, 1, 2) X(3, 2, 2) X(4, 2, 8)

long row_loss_rows(long B) { return (B + ROW_SEQS - 1) / ROW_SEQS; }

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
