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
#include <cstdlib>
#include <mutex>

#include "kalman_core.h"
#include "kalman_quad.h"
#include "launch_common.h"
#include "wave_api.h"

namespace kvae {

constexpr int SEQ_BLOCK = 64;   // threads per workgroup, sequential kernels
constexpr int PAIR_BLOCK = 256; // threads per workgroup, (t,b)-parallel kernels


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

// Quad form (kalman_quad.h): four lanes per sequence, 16 sequences per wavefront.
// Lanes of sequences past the end stay in the kernel (the DPP exchanges need all
// four lanes of a quad), read sequence B-1 and store nothing.
constexpr int QUAD_SEQS = SEQ_BLOCK / 4;

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK) filter_forward_quad_kernel(FilterFwdArgs<R> g) {
    const int b = blockIdx.x * QUAD_SEQS + (threadIdx.x >> 2);
    const bool active = b < g.B;
    quad::filter_forward_quad<R, N, P, K>(g, active ? b : g.B - 1, active);
}

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK) filter_backward_quad_kernel(FilterBwdArgs<R> g) {
    const int b = blockIdx.x * QUAD_SEQS + (threadIdx.x >> 2);
    const bool active = b < g.f.B;
    quad::QFilterAcc<R, N, P, K> acc = {};
    quad::filter_backward_quad<R, N, P, K>(g, active ? b : g.f.B - 1, active, acc);
    quad::write_filter_partial<R, N, P, K>(acc, g.partial, blockIdx.x);
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
__global__ void __launch_bounds__(SEQ_BLOCK) smooth_forward_quad_kernel(SmoothFwdArgs<R> g) {
    const int b = blockIdx.x * QUAD_SEQS + (threadIdx.x >> 2);
    const bool active = b < g.B;
    quad::smooth_forward_quad<R, N, P>(g, active ? b : g.B - 1, active);
}

template <typename R, int N, int P>
__global__ void __launch_bounds__(SEQ_BLOCK) smooth_backward_quad_kernel(SmoothBwdArgs<R> g) {
    constexpr int LEN = P * P;
    const int b = blockIdx.x * QUAD_SEQS + (threadIdx.x >> 2);
    const bool active = b < g.f.B;
    R acc[LEN];     // replicated in the four lanes of a quad
    vzero<R, LEN>(acc);
    quad::smooth_backward_quad<R, N, P>(g, active ? b : g.f.B - 1, active, acc);
    if (g.partial) {
        const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
        for (int e = 0; e < LEN; ++e) {
            const R s = quad::sum_over_quads(acc[e]);
            if (lane == 0) g.partial[(long)blockIdx.x * LEN + e] = s;
        }
    }
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

// ---------------------------------------------------------------------------
// Host side: argument checks, dispatch over the compiled (n,p,K), launches.
// ---------------------------------------------------------------------------
inline long seq_blocks(long B) { return (B + SEQ_BLOCK - 1) / SEQ_BLOCK; }
inline long quad_blocks(long B) { return (B + QUAD_SEQS - 1) / QUAD_SEQS; }

// Four lanes per sequence while one lane per sequence would leave SIMDs without a
// wavefront.  Measured crossover (MI355X, n=4, p=2): the quad kernels are 1.5-2x
// faster up to 16k sequences and slower from 64k on, except the filter forward
// kernel, which is faster at every size.  KVAE_SEQ_LANES=1|4 forces one form
// (tests, timing).
template <int N, int P>
inline bool use_quad(long B, bool at_any_size = false) {
    if (N > 4 || P > 4) return false;
    const char* e = getenv("KVAE_SEQ_LANES");
    const int forced = e ? atoi(e) : 0;
    if (forced == 1) return false;
    if (forced == 4) return true;
    return at_any_size || B < 32768;
}
inline long seq_rows(long B) { return seq_blocks(B) * (SEQ_BLOCK / WAVE); }
inline long pair_blocks(long TB) { return (TB + PAIR_BLOCK - 1) / PAIR_BLOCK; }
inline long pair_rows(long TB) { return pair_blocks(TB) * (PAIR_BLOCK / WAVE); }

// ---- filter ----------------------------------------------------------------
template <typename R, int N, int P, int K>
int run_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s) {
    FilterFwdArgs<R> g = make_args<R>(d, io);
    if constexpr (N <= 4 && P <= 4) {
        if (use_quad<N, P>(d->B, true)) {
            hipLaunchKernelGGL((filter_forward_quad_kernel<R, N, P, K>), dim3(quad_blocks(d->B)),
                               dim3(SEQ_BLOCK), 0, s, g);
            return launch_status();
        }
    }
    hipLaunchKernelGGL((filter_forward_kernel<R, N, P, K>), dim3(seq_blocks(d->B)), dim3(SEQ_BLOCK),
                       0, s, g);
    return launch_status();
}

template <typename R, int N, int P, int K>
int run_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io, void* ws, size_t ws_bytes,
                        hipStream_t s) {
    using PG = FilterParamGrads<N, P, K>;
    FilterBwdArgs<R> g = make_args<R>(d, io, ws);
    if constexpr (N <= 4 && P <= 4) {
        if (use_quad<N, P>(d->B)) {
            const long qrows = quad_blocks(d->B);
            if (ws_bytes < (size_t)qrows * PG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
            hipLaunchKernelGGL((filter_backward_quad_kernel<R, N, P, K>), dim3(qrows),
                               dim3(SEQ_BLOCK), 0, s, g);
            reduce_filter_grads<R, N, P, K>(io, ws, qrows, s);
            return launch_status();
        }
    }
    const long rows = seq_rows(d->B);
    if (ws_bytes < (size_t)rows * PG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
    hipLaunchKernelGGL((filter_backward_kernel<R, N, P, K>), dim3(seq_blocks(d->B)),
                       dim3(SEQ_BLOCK), 0, s, g);
    reduce_filter_grads<R, N, P, K>(io, ws, rows, s);
    return launch_status();
}

// ---- smoother --------------------------------------------------------------
template <typename R, int N, int P>
int run_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s) {
    SmoothFwdArgs<R> g = make_args<R>(d, io);
    if constexpr (N <= 4 && P <= 4) {
        if (use_quad<N, P>(d->B)) {
            hipLaunchKernelGGL((smooth_forward_quad_kernel<R, N, P>), dim3(quad_blocks(d->B)),
                               dim3(SEQ_BLOCK), 0, s, g);
            return launch_status();
        }
    }
    hipLaunchKernelGGL((smooth_forward_kernel<R, N, P>), dim3(seq_blocks(d->B)), dim3(SEQ_BLOCK), 0,
                       s, g);
    return launch_status();
}

template <typename R, int N, int P>
int run_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws, size_t ws_bytes,
                        hipStream_t s) {
    SmoothBwdArgs<R> g = make_args<R>(d, io, ws);
    if constexpr (N <= 4 && P <= 4) {
        if (use_quad<N, P>(d->B)) {
            const long qrows = quad_blocks(d->B);
            if (io->d_LR && ws_bytes < (size_t)qrows * P * P * sizeof(R)) return KVAE_E_WORKSPACE;
            hipLaunchKernelGGL((smooth_backward_quad_kernel<R, N, P>), dim3(qrows), dim3(SEQ_BLOCK),
                               0, s, g);
            if (io->d_LR) reduce_smooth_grads<R, P>(io, ws, qrows, s);
            return launch_status();
        }
    }
    const long rows = seq_rows(d->B);
    if (io->d_LR && ws_bytes < (size_t)rows * P * P * sizeof(R)) return KVAE_E_WORKSPACE;
    hipLaunchKernelGGL((smooth_backward_kernel<R, N, P>), dim3(seq_blocks(d->B)), dim3(SEQ_BLOCK),
                       0, s, g);
    if (io->d_LR) reduce_smooth_grads<R, P>(io, ws, rows, s);
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
    ElboBwdArgs<R> g = make_args<R>(d, io, ws);
    hipLaunchKernelGGL((elbo_backward_kernel<R, N, P>), dim3(pair_blocks((long)d->T * d->B)),
                       dim3(PAIR_BLOCK), 0, s, g);
    reduce_elbo_grads<R, N, P>(io, ws, rows, s);
    return launch_status();
}

}  // namespace kvae

using namespace kvae;

// Expands `CALL(R, n, p, K)` for the compiled combination matching d, else
// returns KVAE_E_UNSUPPORTED.
// The imputation the masked filter does before each call of the weight network
// (kvae/state_space_model.py:232-277), one lane per sequence:
//   z = A(w) m_pred (+ L_Q eps_z),  a_imp = C(w) z (+ L_R eps_a),  a_w = mu a + (1 - mu) a_imp
// with A(w), C(w) the K-way mixtures for this sequence's weights w.  As separate library
// calls this is ~15 launches per step.
template <typename R>
struct ImputeArgs {
    long B;
    const R *w, *A_K, *C_K, *m_pred, *a, *mask, *eps_z, *L_Q, *eps_a, *L_R;
    R* a_w;
};

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK) impute_kernel(ImputeArgs<R> g) {
    const long b = (long)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (b >= g.B) return;
    R w[K], m[N], z[N];
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = g.w[b * K + k];
#pragma unroll
    for (int j = 0; j < N; ++j) m[j] = g.m_pred[b * N + j];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R acc = R(0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            R row = R(0);
#pragma unroll
            for (int j = 0; j < N; ++j) row += g.A_K[(k * N + i) * N + j] * m[j];
            acc += w[k] * row;
        }
        if (g.eps_z) {
#pragma unroll
            for (int j = 0; j < N; ++j) acc += g.L_Q[i * N + j] * g.eps_z[b * N + j];
        }
        z[i] = acc;
    }
    const R mu = g.mask[b];
#pragma unroll
    for (int r = 0; r < P; ++r) {
        R acc = R(0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            R row = R(0);
#pragma unroll
            for (int i = 0; i < N; ++i) row += g.C_K[(k * P + r) * N + i] * z[i];
            acc += w[k] * row;
        }
        if (g.eps_a) {
#pragma unroll
            for (int j = 0; j < P; ++j) acc += g.L_R[r * P + j] * g.eps_a[b * P + j];
        }
        g.a_w[b * P + r] = mu * g.a[b * P + r] + (R(1) - mu) * acc;
    }
}

template <typename R, int N, int P, int K>
int run_impute(const ImputeArgs<R>& g, hipStream_t s) {
    hipLaunchKernelGGL((impute_kernel<R, N, P, K>), dim3((unsigned)seq_blocks(g.B)), dim3(SEQ_BLOCK),
                       0, s, g);
    return launch_status();
}

// Adjoint of that imputation for one time step, one lane per sequence, given the mixed
// matrices A_t, C_t the filter step wrote: from the gradient w.r.t. a_w (the sum of up to two
// tensors) the gradients w.r.t. the observation (added to d_a), the imputed observation
// (d_aimp) and the imputed state (dz) -- the operands of the caller's sums over all steps for
// L_R and L_Q --, the outer products that are the gradients w.r.t. C_t and A_t, and the
// gradient w.r.t. m_pred, which joins what the filter segment of this step returned for its
// initial state: with g_mn_prev / g_Pn_prev (the cotangents of the previous step's predicted
// state) both are added there, without them (first step) d_m_seg takes it.
template <typename R>
struct ImputeBwdArgs {
    long B;
    const R *d_aw_a, *d_aw_b, *mask, *A_t, *C_t, *m_pred, *eps_z, *L_Q;
    R *d_a, *d_aimp, *dz, *gC, *gA, *d_m_seg;
    const R* d_P_seg;
    R *g_mn_prev, *g_Pn_prev;
};

template <typename R, int N, int P>
__global__ void __launch_bounds__(SEQ_BLOCK) impute_backward_kernel(ImputeBwdArgs<R> g) {
    const long b = (long)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (b >= g.B) return;
    R m[N], z[N], dz[N], da[P];
    const R mu = g.mask[b];
#pragma unroll
    for (int j = 0; j < N; ++j) m[j] = g.m_pred[b * N + j];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R acc = R(0);
#pragma unroll
        for (int j = 0; j < N; ++j) acc += g.A_t[(b * N + i) * N + j] * m[j];
        if (g.eps_z) {
#pragma unroll
            for (int j = 0; j < N; ++j) acc += g.L_Q[i * N + j] * g.eps_z[b * N + j];
        }
        z[i] = acc;
        dz[i] = R(0);
    }
#pragma unroll
    for (int r = 0; r < P; ++r) {
        const R d = g.d_aw_a[b * P + r] + (g.d_aw_b ? g.d_aw_b[b * P + r] : R(0));
        g.d_a[b * P + r] += mu * d;
        da[r] = (R(1) - mu) * d;
        g.d_aimp[b * P + r] = da[r];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            dz[i] += g.C_t[(b * P + r) * N + i] * da[r];
            g.gC[(b * P + r) * N + i] = da[r] * z[i];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        g.dz[b * N + i] = dz[i];
#pragma unroll
        for (int j = 0; j < N; ++j) g.gA[(b * N + i) * N + j] = dz[i] * m[j];
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        R back = R(0);
#pragma unroll
        for (int i = 0; i < N; ++i) back += g.A_t[(b * N + i) * N + j] * dz[i];
        if (g.g_mn_prev)
            g.g_mn_prev[b * N + j] += g.d_m_seg[b * N + j] + back;
        else
            g.d_m_seg[b * N + j] += back;
    }
    if (g.g_Pn_prev) {
#pragma unroll
        for (int e = 0; e < N * N; ++e) g.g_Pn_prev[b * N * N + e] += g.d_P_seg[b * N * N + e];
    }
}

template <typename R, int N, int P>
int run_impute_backward(const ImputeBwdArgs<R>& g, hipStream_t s) {
    hipLaunchKernelGGL((impute_backward_kernel<R, N, P>), dim3((unsigned)seq_blocks(g.B)),
                       dim3(SEQ_BLOCK), 0, s, g);
    return launch_status();
}

// Masked filter with the two-layer perceptron weight network evaluated in the loop
// (filter_forward_masked_mlp_sequence): the whole T-step recursion of masked sequences
// is ONE launch, where stepping the network between filter steps takes three launches
// per time step.  Every lane reads the same entry of the packed network (a few KB) at the
// same time: `net` is a read-only, non-aliased kernel argument so that those loads are
// scalar loads (one per wavefront, served by the scalar cache) feeding SGPR operands.
constexpr int MASKED_NET_MAX_BYTES = 64 * 1024;   // read through the scalar cache; not staged

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK)
filter_forward_masked_mlp_kernel(FilterFwdArgs<R> g, MaskedNetArgs<R> q,
                                 const R* __restrict__ net) {
    q.net = net;
    const int b = blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (b < g.B) filter_forward_masked_mlp_sequence<R, N, P, K>(g, q, b);
}

template <typename R, int N, int P, int K>
int run_filter_forward_masked_mlp(const kvae_dims* d, const kvae_filter_fwd* io,
                                  const kvae_masked_mlp* net, hipStream_t s) {
    if ((size_t)(net->hidden * (P + 1 + K) + K) * sizeof(R) > (size_t)MASKED_NET_MAX_BYTES)
        return KVAE_E_UNSUPPORTED;
    FilterFwdArgs<R> g = make_args<R>(d, io);
    MaskedNetArgs<R> q{net->hidden,          nullptr,               (const R*)net->eps_z,
                       (const R*)net->L_Q,   (const R*)net->eps_a,  (const R*)net->L_R,
                       (R*)net->a_w,         (R*)io->alpha};
    hipLaunchKernelGGL((filter_forward_masked_mlp_kernel<R, N, P, K>), dim3(seq_blocks(d->B)),
                       dim3(SEQ_BLOCK), 0, s, g, q, (const R*)net->packed);
    return launch_status();
}

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
        case KVAE_E_TOOLARGE:
            return "problem needs more scratch than the no-allocation entry point holds";
        default: return "unknown error";
    }
}

int kvae_supported(int n, int p, int K, int dtype) {
    if (dtype != KVAE_F32 && dtype != KVAE_F64) return 0;
#define KVAE_SUP(n_, p_, K_) \
    if (n == n_ && p == p_ && K == K_) return 1;
    KVAE_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return wave_supported(n, p, K, dtype);
}

// 1 when the lane-per-sequence kernels cover (n, p, K)
static int lane_supported(int n, int p, int K) {
#define KVAE_SUP(n_, p_, K_) \
    if (n == n_ && p == p_ && K == K_) return 1;
    KVAE_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return 0;
}

size_t kvae_workspace_bytes(const kvae_dims* d, int op) {
    if (!dims_ok(d)) return 0;
    if (!lane_supported(d->n, d->p, d->K)) return wave_workspace_bytes(d, op);
    const size_t el = d->dtype == KVAE_F32 ? 4 : 8;
    const long n = d->n, p = d->p, K = d->K;
    switch (op) {
        case KVAE_OP_FILTER_BWD:   // one row per wavefront; the quad form has 16 sequences in one
            return (size_t)(n <= 4 && p <= 4 ? quad_blocks(d->B) : seq_rows(d->B)) *
                   (K * n * n + K * p * n + n * n + p * p) * el;
        case KVAE_OP_SMOOTH_BWD:
            return (size_t)(n <= 4 && p <= 4 ? quad_blocks(d->B) : seq_rows(d->B)) * (p * p) * el;
        case KVAE_OP_ELBO_BWD:
            return (size_t)pair_rows((long)d->T * d->B) * (n * n + p * p + n + n * n) * el;
        case KVAE_OP_LOSS_FWD:
        case KVAE_OP_LOSS_BWD:
        case KVAE_OP_LOSS_VALUE_AND_GRAD:
            return loss_workspace_bytes(d, op);
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
    return wave_filter_forward(d, io, (hipStream_t)stream);
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
    return wave_filter_backward(d, io, workspace, workspace_bytes, (hipStream_t)stream);
}

// kvae_filter_backward for callers that cannot allocate: the per-wavefront partial sums of
// the parameter gradients go to this array of the code object instead of a caller-provided
// workspace.  Every GPU the library runs on holds its own copy at its own address; host
// code may only use the address hipGetSymbolAddress returns on that device (the array's
// name is a host-side stand-in, not a device address).
static __device__ __align__(256) unsigned char scratch[KVAE_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex scratch_lock;
void* scratch_address[MAX_DEVICES] = {};

// Device address of `scratch` on the current device (looked up once per device).
int static_scratch(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return KVAE_E_LAUNCH;
    std::lock_guard<std::mutex> hold(scratch_lock);
    if (!scratch_address[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return KVAE_E_LAUNCH;
        scratch_address[dev] = p;
    }
    *out = scratch_address[dev];
    return KVAE_OK;
}
}  // namespace

size_t kvae_noalloc_scratch_bytes(void) { return KVAE_NOALLOC_SCRATCH_BYTES; }

int kvae_filter_backward_noalloc(const kvae_dims* d, const kvae_filter_bwd* io, void* stream) {
    if (!dims_ok(d)) return KVAE_E_BADARG;
    const size_t need = kvae_workspace_bytes(d, KVAE_OP_FILTER_BWD);
    if (need == 0) return KVAE_E_UNSUPPORTED;
    if (need > KVAE_NOALLOC_SCRATCH_BYTES) return KVAE_E_TOOLARGE;
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != KVAE_OK) return rc;
    // every wavefront overwrites its own row of partial sums and the row-sum kernel reads
    // only the rows of this call: nothing of an earlier call has to be cleared
    return kvae_filter_backward(d, io, ws, need, stream);
}

int kvae_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, void* stream) {
    if (!dims_ok(d) || !io || !io->mean_filt || !io->cov_filt || !io->mean_pred ||
        !io->cov_pred || !io->mat_As || !io->mat_Cs || !io->mean_smooth || !io->cov_smooth ||
        !io->zs || !io->as_resampled || (io->eps_obs && !io->LR))
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) run_smooth_forward<R, n_, p_>(d, io, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return wave_smooth_forward(d, io, (hipStream_t)stream);
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
    return wave_smooth_backward(d, io, workspace, workspace_bytes, (hipStream_t)stream);
}

int kvae_filter_smooth_forward(const kvae_dims* d, const kvae_filter_fwd* f,
                               const kvae_smooth_fwd* s, void* stream) {
    if (!dims_ok(d) || !f || !s || !f->mat_As || !f->mat_Cs || s->mean_filt != f->mean_filt ||
        s->cov_filt != f->cov_filt || s->mean_pred != f->mean_pred ||
        s->cov_pred != f->cov_pred || s->mat_As != f->mat_As || s->mat_Cs != f->mat_Cs)
        return KVAE_E_BADARG;
    // From 32768 sequences on (where the filter and the smoother would both run one lane per
    // sequence anyway) the two sweeps are ONE launch: the reverse sweep re-mixes the per-step
    // matrices and recomputes the predictions instead of reading back what the forward sweep
    // wrote.  Below that the two four-lanes-per-sequence kernels are faster than one launch.
    const char* e = getenv("KVAE_SEQ_LANES");
    const int forced = e ? atoi(e) : 0;
    if ((d->B >= 32768 && forced != 4) || forced == 1) {
        const int rc1 = kvae_filter_smooth_fused(d, f, nullptr, s, nullptr, 0, stream);
        if (rc1 != KVAE_E_UNSUPPORTED) return rc1;
    }
    const int rc = kvae_filter_forward(d, f, stream);
    if (rc != KVAE_OK) return rc;
    return kvae_smooth_forward(d, s, stream);
}

int kvae_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, void* stream) {
    if (!dims_ok(d) || !io || !io->a || !io->mean_smooth || !io->cov_smooth || !io->mat_As ||
        !io->mat_Cs || !io->Q || !io->R || !io->m0 || !io->P0 || !io->eps || !io->obs ||
        !io->trans || !io->post)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) run_elbo_forward<R, n_, p_>(d, io, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return wave_elbo_forward(d, io, (hipStream_t)stream);
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
    return wave_elbo_backward(d, io, workspace, workspace_bytes, (hipStream_t)stream);
}

int kvae_impute_observation(const kvae_dims* d, const void* w, const void* A_K, const void* C_K,
                            const void* m_pred, const void* a, const void* mask, const void* eps_z,
                            const void* L_Q, const void* eps_a, const void* L_R, void* a_w,
                            void* stream) {
    if (!dims_ok(d) || !w || !A_K || !C_K || !m_pred || !a || !mask || !a_w || (eps_z && !L_Q) ||
        (eps_a && !L_R))
        return KVAE_E_BADARG;
#define KVAE_IMPUTE(R)                                                                         \
    ImputeArgs<R> g{d->B,          (const R*)w,     (const R*)A_K,  (const R*)C_K,             \
                    (const R*)m_pred, (const R*)a,  (const R*)mask, (const R*)eps_z,           \
                    (const R*)L_Q, (const R*)eps_a, (const R*)L_R,  (R*)a_w}
#define KVAE_CALL(R, n_, p_, K_) [&] { KVAE_IMPUTE(R); return run_impute<R, n_, p_, K_>(g, (hipStream_t)stream); }()
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
#undef KVAE_IMPUTE
    return KVAE_E_UNSUPPORTED;
}

int kvae_impute_observation_backward(const kvae_dims* d, const void* d_aw_a, const void* d_aw_b,
                                     const void* mask, const void* A_t, const void* C_t,
                                     const void* m_pred, const void* eps_z, const void* L_Q,
                                     void* d_a, void* d_aimp, void* dz, void* gC, void* gA,
                                     void* d_m_seg, const void* d_P_seg, void* g_mn_prev,
                                     void* g_Pn_prev, void* stream) {
    if (!dims_ok(d) || !d_aw_a || !mask || !A_t || !C_t || !m_pred || (eps_z && !L_Q) || !d_a ||
        !d_aimp || !dz || !gC || !gA || !d_m_seg || (g_Pn_prev && !d_P_seg) ||
        (g_mn_prev == nullptr) != (g_Pn_prev == nullptr))
        return KVAE_E_BADARG;
#define KVAE_IMPUTE_B(R)                                                                        \
    ImputeBwdArgs<R> g{d->B,           (const R*)d_aw_a, (const R*)d_aw_b, (const R*)mask,      \
                       (const R*)A_t,  (const R*)C_t,    (const R*)m_pred, (const R*)eps_z,     \
                       (const R*)L_Q,  (R*)d_a,          (R*)d_aimp,       (R*)dz,              \
                       (R*)gC,         (R*)gA,           (R*)d_m_seg,      (const R*)d_P_seg,   \
                       (R*)g_mn_prev,  (R*)g_Pn_prev}
#define KVAE_CALL(R, n_, p_, K_) [&] { KVAE_IMPUTE_B(R); return run_impute_backward<R, n_, p_>(g, (hipStream_t)stream); }()
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
#undef KVAE_IMPUTE_B
    return KVAE_E_UNSUPPORTED;
}

int kvae_filter_forward_masked_mlp(const kvae_dims* d, const kvae_filter_fwd* io,
                                   const kvae_masked_mlp* net, void* stream) {
    if (!dims_ok(d) || !io || !io->a || !io->alpha || !io->A_K || !io->C_K || !io->Q || !io->R ||
        !io->m_in || !io->P_in || !io->mean_filt || !io->cov_filt || !io->mean_pred ||
        !io->cov_pred || !net || net->hidden < 1 || !net->packed || !net->a_w ||
        (net->eps_z && !net->L_Q) || (net->eps_a && !net->L_R))
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_filter_forward_masked_mlp<R, n_, p_, K_>(d, io, net, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

}  // extern "C"
