// gfx950 kernels and the C ABI (include/kvae_kalman.h) of the lane-per-sequence
// Kalman filter / RTS smoother / Kalman-ELBO path.
//
// Launch shape: one wavefront (64 lanes) per workgroup for the sequential
// kernels, lane = sequence; the ELBO kernels map one lane to one (t,b) pair.
// Many small workgroups spread a modest batch over all 256 CUs (B = 4096 is
// only 64 wavefronts) and the kernels use no LDS except the final parameter-
// gradient reduction, which is done with wavefront shuffles.
//
// Parameter gradients (sums over T and B) are reduced in two deterministic
// stages: lane accumulators -> one row of partial sums per wavefront
// (shuffle reduction) -> reduce_rows_kernel sums the rows.  No float atomics,
// so results are bitwise reproducible run to run.
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "kalman_core.h"

namespace kvae {

constexpr int WAVE = 64;
constexpr int SEQ_BLOCK = 64;   // threads per workgroup, sequential kernels
constexpr int PAIR_BLOCK = 256; // threads per workgroup, (t,b)-parallel kernels

template <typename R>
__device__ __forceinline__ R wave_sum(R v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// Sum acc[0..LEN) over the wavefront and let lane 0 write row `row`.
template <typename R, int LEN>
__device__ __forceinline__ void write_wave_partial(R* acc, R* partial, long row) {
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int e = 0; e < LEN; ++e) {
        const R s = wave_sum<R>(acc[e]);
        if (lane == 0) partial[row * LEN + e] = s;
    }
}

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK) filter_forward_kernel(FilterFwdArgs<R> g) {
    const int b = blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (b < g.B) filter_forward_sequence<R, N, P, K>(g, b);
}

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK) filter_backward_kernel(FilterBwdArgs<R> g) {
    constexpr int LEN = FilterParamGrads<N, P, K>::LEN;
    const int b = blockIdx.x * SEQ_BLOCK + threadIdx.x;
    R acc[LEN];
    vzero<R, LEN>(acc);
    if (b < g.f.B) filter_backward_sequence<R, N, P, K>(g, b, acc);
    const long row = (long)blockIdx.x * (SEQ_BLOCK / WAVE) + (threadIdx.x / WAVE);
    write_wave_partial<R, LEN>(acc, g.partial, row);
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(SEQ_BLOCK) smooth_forward_kernel(SmoothFwdArgs<R> g) {
    const int b = blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (b < g.B) smooth_forward_sequence<R, N, P>(g, b);
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(SEQ_BLOCK) smooth_backward_kernel(SmoothBwdArgs<R> g) {
    constexpr int LEN = P * P;
    const int b = blockIdx.x * SEQ_BLOCK + threadIdx.x;
    R acc[LEN];
    vzero<R, LEN>(acc);
    if (b < g.f.B) smooth_backward_sequence<R, N, P>(g, b, acc);
    if (g.partial) {
        const long row = (long)blockIdx.x * (SEQ_BLOCK / WAVE) + (threadIdx.x / WAVE);
        write_wave_partial<R, LEN>(acc, g.partial, row);
    }
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(PAIR_BLOCK) elbo_forward_kernel(ElboFwdArgs<R> g) {
    const long i = (long)blockIdx.x * PAIR_BLOCK + threadIdx.x;
    if (i < (long)g.T * g.B) elbo_forward_pair<R, N, P>(g, (int)(i / g.B), (int)(i % g.B));
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(PAIR_BLOCK) elbo_backward_kernel(ElboBwdArgs<R> g) {
    constexpr int LEN = ElboParamGrads<N, P>::LEN;
    const long i = (long)blockIdx.x * PAIR_BLOCK + threadIdx.x;
    R acc[LEN];
    vzero<R, LEN>(acc);
    if (i < (long)g.f.T * g.f.B)
        elbo_backward_pair<R, N, P>(g, (int)(i / g.f.B), (int)(i % g.f.B), acc);
    const long row = (long)blockIdx.x * (PAIR_BLOCK / WAVE) + (threadIdx.x / WAVE);
    write_wave_partial<R, LEN>(acc, g.partial, row);
}

// out[c] = sum_r partial[r][c]; one workgroup per column, fixed summation
// order.  Columns are routed to up to five output tensors by offset.
struct ReduceSegments {
    void* out[5];
    int begin[6];  // begin[i]..begin[i+1] = columns of out[i]
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

// ---------------------------------------------------------------------------
// Host side: argument checks, dispatch over the compiled (n,p,K), launches.
// ---------------------------------------------------------------------------
#define KVAE_DIM_LIST(X) \
    X(4, 2, 3)           \
    X(2, 1, 2)           \
    X(3, 2, 2)           \
    X(8, 4, 4)

inline long seq_blocks(long B) { return (B + SEQ_BLOCK - 1) / SEQ_BLOCK; }
inline long seq_rows(long B) { return seq_blocks(B) * (SEQ_BLOCK / WAVE); }
inline long pair_blocks(long TB) { return (TB + PAIR_BLOCK - 1) / PAIR_BLOCK; }
inline long pair_rows(long TB) { return pair_blocks(TB) * (PAIR_BLOCK / WAVE); }

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

// ---- filter ----------------------------------------------------------------
template <typename R, int N, int P, int K>
int run_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s) {
    FilterFwdArgs<R> g = make_args<R>(d, io);
    hipLaunchKernelGGL((filter_forward_kernel<R, N, P, K>), dim3(seq_blocks(d->B)), dim3(SEQ_BLOCK),
                       0, s, g);
    return launch_status();
}

template <typename R, int N, int P, int K>
int run_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io, void* ws, size_t ws_bytes,
                        hipStream_t s) {
    using PG = FilterParamGrads<N, P, K>;
    const long rows = seq_rows(d->B);
    if (ws_bytes < (size_t)rows * PG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
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
    hipLaunchKernelGGL((filter_backward_kernel<R, N, P, K>), dim3(seq_blocks(d->B)),
                       dim3(SEQ_BLOCK), 0, s, g);
    ReduceSegments seg;
    seg.count = 4;
    seg.out[0] = io->d_A_K;
    seg.out[1] = io->d_C_K;
    seg.out[2] = io->d_Q;
    seg.out[3] = io->d_R;
    seg.out[4] = nullptr;
    seg.begin[0] = PG::OFF_A;
    seg.begin[1] = PG::OFF_C;
    seg.begin[2] = PG::OFF_Q;
    seg.begin[3] = PG::OFF_R;
    seg.begin[4] = PG::LEN;
    seg.begin[5] = PG::LEN;
    hipLaunchKernelGGL((reduce_rows_kernel<R>), dim3(PG::LEN), dim3(256), 0, s,
                       (const R*)ws, rows, (int)PG::LEN, seg);
    return launch_status();
}

// ---- smoother --------------------------------------------------------------
template <typename R, int N, int P>
int run_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s) {
    SmoothFwdArgs<R> g = make_args<R>(d, io);
    hipLaunchKernelGGL((smooth_forward_kernel<R, N, P>), dim3(seq_blocks(d->B)), dim3(SEQ_BLOCK), 0,
                       s, g);
    return launch_status();
}

template <typename R, int N, int P>
int run_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws, size_t ws_bytes,
                        hipStream_t s) {
    const long rows = seq_rows(d->B);
    const bool want_LR = io->d_LR != nullptr;
    if (want_LR && ws_bytes < (size_t)rows * P * P * sizeof(R)) return KVAE_E_WORKSPACE;
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
    g.partial = want_LR ? (R*)ws : nullptr;
    hipLaunchKernelGGL((smooth_backward_kernel<R, N, P>), dim3(seq_blocks(d->B)), dim3(SEQ_BLOCK),
                       0, s, g);
    if (want_LR) {
        ReduceSegments seg;
        seg.count = 1;
        for (int i = 0; i < 5; ++i) seg.out[i] = nullptr;
        seg.out[0] = io->d_LR;
        seg.begin[0] = 0;
        for (int i = 1; i < 6; ++i) seg.begin[i] = P * P;
        hipLaunchKernelGGL((reduce_rows_kernel<R>), dim3(P * P), dim3(256), 0, s, (const R*)ws,
                           rows, P * P, seg);
    }
    return launch_status();
}

// ---- ELBO ------------------------------------------------------------------
template <typename R, int N, int P>
int run_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, hipStream_t s) {
    ElboFwdArgs<R> g = make_args<R>(d, io);
    hipLaunchKernelGGL((elbo_forward_kernel<R, N, P>), dim3(pair_blocks((long)d->T * d->B)),
                       dim3(PAIR_BLOCK), 0, s, g);
    return launch_status();
}

template <typename R, int N, int P>
int run_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws, size_t ws_bytes,
                      hipStream_t s) {
    using PG = ElboParamGrads<N, P>;
    const long rows = pair_rows((long)d->T * d->B);
    if (ws_bytes < (size_t)rows * PG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
    ElboBwdArgs<R> g;
    g.f = make_args<R>(d, &io->fwd);
    g.g_obs = (const R*)io->g_obs;
    g.g_trans = (const R*)io->g_trans;
    g.g_post = (const R*)io->g_post;
    g.d_a = (R*)io->d_a;
    g.d_ms = (R*)io->d_mean_smooth;
    g.d_Ps = (R*)io->d_cov_smooth;
    g.d_As = (R*)io->d_mat_As;
    g.d_Cs = (R*)io->d_mat_Cs;
    g.partial = (R*)ws;
    hipLaunchKernelGGL((elbo_backward_kernel<R, N, P>), dim3(pair_blocks((long)d->T * d->B)),
                       dim3(PAIR_BLOCK), 0, s, g);
    ReduceSegments seg;
    seg.count = 4;
    seg.out[0] = io->d_Q;
    seg.out[1] = io->d_R;
    seg.out[2] = io->d_m0;
    seg.out[3] = io->d_P0;
    seg.out[4] = nullptr;
    seg.begin[0] = PG::OFF_Q;
    seg.begin[1] = PG::OFF_R;
    seg.begin[2] = PG::OFF_M0;
    seg.begin[3] = PG::OFF_P0;
    seg.begin[4] = PG::LEN;
    seg.begin[5] = PG::LEN;
    hipLaunchKernelGGL((reduce_rows_kernel<R>), dim3(PG::LEN), dim3(256), 0, s, (const R*)ws, rows,
                       (int)PG::LEN, seg);
    return launch_status();
}

}  // namespace kvae

using namespace kvae;

// Expands `CALL(R, n, p, K)` for the compiled combination matching d, else
// returns KVAE_E_UNSUPPORTED.
#define KVAE_DISPATCH_CASE(n_, p_, K_)                                  \
    if (d->n == n_ && d->p == p_ && d->K == K_) {                       \
        if (d->dtype == KVAE_F32) return KVAE_CALL(float, n_, p_, K_);  \
        return KVAE_CALL(double, n_, p_, K_);                           \
    }

extern "C" {

int kvae_version(void) { return KVAE_ABI_VERSION; }

const char* kvae_error_string(int code) {
    switch (code) {
        case KVAE_OK: return "ok";
        case KVAE_E_BADARG: return "bad argument (null pointer or non-positive size)";
        case KVAE_E_UNSUPPORTED: return "no kernel compiled for this (n, p, K, dtype)";
        case KVAE_E_WORKSPACE: return "workspace too small, see kvae_workspace_bytes()";
        case KVAE_E_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

int kvae_supported(int n, int p, int K, int dtype) {
    if (dtype != KVAE_F32 && dtype != KVAE_F64) return 0;
#define KVAE_SUP(n_, p_, K_) \
    if (n == n_ && p == p_ && K == K_) return 1;
    KVAE_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return 0;
}

size_t kvae_workspace_bytes(const kvae_dims* d, int op) {
    if (!dims_ok(d)) return 0;
    const size_t el = d->dtype == KVAE_F32 ? 4 : 8;
    const long n = d->n, p = d->p, K = d->K;
    switch (op) {
        case KVAE_OP_FILTER_BWD:
            return (size_t)seq_rows(d->B) * (K * n * n + K * p * n + n * n + p * p) * el;
        case KVAE_OP_SMOOTH_BWD:
            return (size_t)seq_rows(d->B) * (p * p) * el;
        case KVAE_OP_ELBO_BWD:
            return (size_t)pair_rows((long)d->T * d->B) * (n * n + p * p + n + n * n) * el;
        default:
            return 0;
    }
}

int kvae_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, void* stream) {
    if (!dims_ok(d) || !io || !io->a || !io->alpha || !io->A_K || !io->C_K || !io->Q || !io->R ||
        !io->m_in || !io->P_in || !io->mean_filt || !io->cov_filt || !io->mean_pred ||
        !io->cov_pred)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) run_filter_forward<R, n_, p_, K_>(d, io, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io, void* workspace,
                         size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !io->fwd.a || !io->fwd.alpha || !io->fwd.A_K || !io->fwd.C_K ||
        !io->fwd.Q || !io->fwd.R || !io->fwd.m_in || !io->fwd.P_in || !io->fwd.mean_filt ||
        !io->fwd.cov_filt || !io->fwd.mean_pred || !io->fwd.cov_pred || !io->d_a ||
        !io->d_alpha || !io->d_m_in || !io->d_P_in || !workspace)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_filter_backward<R, n_, p_, K_>(d, io, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, void* stream) {
    if (!dims_ok(d) || !io || !io->mean_filt || !io->cov_filt || !io->mean_pred ||
        !io->cov_pred || !io->mat_As || !io->mat_Cs || !io->mean_smooth || !io->cov_smooth ||
        !io->zs || !io->as_resampled || (io->eps_obs && !io->LR))
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) run_smooth_forward<R, n_, p_>(d, io, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* workspace,
                         size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !io->fwd.mean_filt || !io->fwd.cov_filt || !io->fwd.mean_pred ||
        !io->fwd.cov_pred || !io->fwd.mat_As || !io->fwd.mat_Cs || !io->fwd.mean_smooth ||
        !io->fwd.cov_smooth || (io->fwd.eps_obs && !io->fwd.LR) || !io->d_mean_filt ||
        !io->d_cov_filt || !io->d_mean_pred || !io->d_cov_pred || !io->d_mat_As ||
        !io->d_mat_Cs || (io->d_LR && !workspace))
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_smooth_backward<R, n_, p_>(d, io, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, void* stream) {
    if (!dims_ok(d) || !io || !io->a || !io->mean_smooth || !io->cov_smooth || !io->mat_As ||
        !io->mat_Cs || !io->Q || !io->R || !io->m0 || !io->P0 || !io->eps || !io->obs ||
        !io->trans || !io->post)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) run_elbo_forward<R, n_, p_>(d, io, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !io->fwd.a || !io->fwd.mean_smooth || !io->fwd.cov_smooth ||
        !io->fwd.mat_As || !io->fwd.mat_Cs || !io->fwd.Q || !io->fwd.R || !io->fwd.m0 ||
        !io->fwd.P0 || !io->fwd.eps || !io->d_a || !io->d_mean_smooth || !io->d_cov_smooth ||
        !io->d_mat_As || !io->d_mat_Cs || !workspace)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_elbo_backward<R, n_, p_>(d, io, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

}  // extern "C"
