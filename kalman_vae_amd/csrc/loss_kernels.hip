// gfx950 kernels and C ABI of the fused Kalman objective (kalman_loss.h):
// kvae_kalman_loss_forward / kvae_kalman_loss_backward of include/kvae_kalman.h.
//
// One launch per direction runs filter, RTS smoother and the three log-density
// terms (forward) or their complete adjoint (backward) for every sequence, plus one
// small deterministic row-sum launch each: the forward sums the per-wavefront term
// totals into the four output scalars, the backward sums the per-wavefront partial
// parameter gradients.  No float atomics.
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "kalman_core.h"
#include <cstdlib>

#include "kalman_loss.h"
#include "kalman_loss_quad.h"
#include "launch_common.h"
#include "wave_api.h"

namespace kvae {

constexpr int LOSS_BLOCK = 64;   // one wavefront per workgroup

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(LOSS_BLOCK) loss_forward_kernel(LossFwdArgs<R> g) {
    const int b = blockIdx.x * LOSS_BLOCK + threadIdx.x;
    R sums[3] = {R(0), R(0), R(0)};
    if (b < g.B) loss_forward_sequence<R, N, P, K>(g, b, sums);
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const R s = wave_sum<R>(sums[e]);
        if (lane == 0) g.partial[(long)blockIdx.x * 3 + e] = s;
    }
}

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(LOSS_BLOCK) loss_backward_kernel(LossBwdArgs<R> g) {
    constexpr int LEN = LossParamGrads<N, P, K>::LEN;
    const int b = blockIdx.x * LOSS_BLOCK + threadIdx.x;
    R acc[LEN];
    vzero<R, LEN>(acc);
    if (b < g.f.B) loss_backward_sequence<R, N, P, K>(g, b, acc);
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int e = 0; e < LEN; ++e) {
        const R s = wave_sum<R>(acc[e]);
        if (lane == 0) g.partial[(long)blockIdx.x * LEN + e] = s;
    }
}

// Quad form (kalman_loss_quad.h): four lanes per sequence, 16 sequences per wavefront.
// Lanes of sequences past the end stay in the kernel (the quad exchanges need all four
// lanes), read sequence B-1, store nothing and add nothing to the sums.
constexpr int LOSS_QUAD_SEQS = LOSS_BLOCK / 4;

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(LOSS_BLOCK) loss_forward_quad_kernel(LossFwdArgs<R> g) {
    const int b = blockIdx.x * LOSS_QUAD_SEQS + (threadIdx.x >> 2);
    const bool active = b < g.B;
    R sums[3];
    quad::loss_forward_quad<R, N, P, K>(g, active ? b : g.B - 1, active, sums);
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const R s = quad::sum_over_quads(sums[e]);
        if (lane == 0) g.partial[(long)blockIdx.x * 3 + e] = s;
    }
}

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(LOSS_BLOCK) loss_backward_quad_kernel(LossBwdArgs<R> g) {
    const int b = blockIdx.x * LOSS_QUAD_SEQS + (threadIdx.x >> 2);
    const bool active = b < g.f.B;
    quad::QLossAcc<R, N, P, K> acc = {};
    quad::loss_backward_quad<R, N, P, K>(g, active ? b : g.f.B - 1, active, acc);
    quad::write_loss_partial<R, N, P, K>(acc, g.partial, blockIdx.x);
}

// out (4) = [ w0 * sum_r partial[r][0], w1 * ..., w2 * ..., their total ], fixed order.
template <typename R>
__global__ void __launch_bounds__(256) loss_finish_kernel(const R* partial, long rows, R w0, R w1,
                                                         R w2, R* out) {
    __shared__ R sm[3][256 / WAVE];
    R s[3] = {R(0), R(0), R(0)};
    for (long r = threadIdx.x; r < rows; r += 256) {
#pragma unroll
        for (int e = 0; e < 3; ++e) s[e] += partial[r * 3 + e];
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const R v = wave_sum<R>(s[e]);
        if ((threadIdx.x & (WAVE - 1)) == 0) sm[e][threadIdx.x / WAVE] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        R tot[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            tot[e] = R(0);
            for (int w = 0; w < 256 / WAVE; ++w) tot[e] += sm[e][w];
        }
        out[0] = w0 * tot[0];
        out[1] = w1 * tot[1];
        out[2] = w2 * tot[2];
        out[3] = (out[0] + out[1]) + out[2];
    }
}

// Which form runs a call: 16 lanes per sequence (row form, loss_row_kernels.hip) while all of
// its workgroups are resident at once, four lanes per sequence (quad form) while one lane per
// sequence would leave SIMDs without a wavefront (the chip has 1024), else one lane per
// sequence.  The row form's workgroups hold their sequences' per-step state in LDS, which
// bounds how many a compute unit takes (four at T = 20: 4096 sequences on 256 compute units);
// a second round of workgroups doubles its time, and from there the quad form is faster
// (T = 20: 75 / 160 us at 4096 / 5120 sequences against 131 / 136 us; T = 8: 49 / 81 us at
// 8192 / 12288 against 63 / 68 us).  KVAE_SEQ_LANES=1|4|16 forces a form where it exists
// (tests, timing).  The row form needs its per-step state to fit 64 KB of LDS (T <= 35 with
// the gradient) and exists for float only; its forward call writes nothing into `saved` and its
// backward call reads nothing from it, so both calls of a pair must take the same decision:
// it depends on the dimensions, the environment and the current device only.
inline int loss_form(const kvae_dims* d) {
    const char* e = getenv("KVAE_SEQ_LANES");
    const int forced = e ? atoi(e) : 0;
    const bool quad_ok = d->n <= 4 && d->p <= 4;
    const bool row_ok = row_loss_supported(d->n, d->p, d->K, d->dtype, d->T, 1) != 0;
    if (forced == 16 && row_ok) return 16;
    if (forced == 4 && quad_ok) return 4;
    if (forced == 1) return 1;
    if (forced == 0 && row_ok && d->B <= row_loss_resident_sequences(d->T)) return 16;
    if (quad_ok && d->B < 32768) return 4;
    return 1;
}
inline bool loss_use_quad(const kvae_dims* d) { return loss_form(d) == 4; }
// wavefronts (= rows of partial sums) of a launch of the lane / quad forms
inline long loss_rows(const kvae_dims* d) {
    const long per_wave = loss_use_quad(d) ? LOSS_QUAD_SEQS : LOSS_BLOCK;
    return (d->B + per_wave - 1) / per_wave;
}
// the larger of the two forms, for workspace sizing that does not depend on the environment
inline long loss_rows_max(int n, int p, long B) {
    const long per_wave = (n <= 4 && p <= 4) ? LOSS_QUAD_SEQS : LOSS_BLOCK;
    return (B + per_wave - 1) / per_wave;
}

template <typename R>
LossFwdArgs<R> make_loss_args(const kvae_dims* d, const kvae_loss_fwd* io, void* partial) {
    LossFwdArgs<R> g;
    g.T = d->T;
    g.B = d->B;
    g.symmetrize = d->symmetrize;
    g.a = (const R*)io->a;
    g.alpha = (const R*)io->alpha;
    g.mask = (const R*)io->mask;
    g.A_K = (const R*)io->A_K;
    g.C_K = (const R*)io->C_K;
    g.Q = (const R*)io->Q;
    g.Rm = (const R*)io->R;
    g.m0 = (const R*)io->m0;
    g.P0 = (const R*)io->P0;
    g.eps = (const R*)io->eps;
    loss_saved_pointers<R>(io->saved, d->T, d->B, d->n, g);
    g.terms = (R*)io->terms;
    g.partial = (R*)partial;
    g.status = io->status;
    return g;
}

template <typename R, int N, int P, int K>
int run_loss_forward(const kvae_dims* d, const kvae_loss_fwd* io, void* ws, size_t ws_bytes,
                     hipStream_t s) {
    if (loss_form(d) == 16) return row_loss_forward(d, io, ws, ws_bytes, s);
    const long rows = loss_rows(d);
    if (ws_bytes < (size_t)rows * 3 * sizeof(R)) return KVAE_E_WORKSPACE;
    LossFwdArgs<R> g = make_loss_args<R>(d, io, ws);
    bool launched = false;
    if constexpr (N <= 4 && P <= 4) {
        if (loss_use_quad(d)) {
            hipLaunchKernelGGL((loss_forward_quad_kernel<R, N, P, K>), dim3(rows),
                               dim3(LOSS_BLOCK), 0, s, g);
            launched = true;
        }
    }
    if (!launched)
        hipLaunchKernelGGL((loss_forward_kernel<R, N, P, K>), dim3(rows), dim3(LOSS_BLOCK), 0, s,
                           g);
    hipLaunchKernelGGL((loss_finish_kernel<R>), dim3(1), dim3(256), 0, s, (const R*)ws, rows,
                       (R)io->w_obs, (R)io->w_trans, (R)io->w_post, (R*)io->out);
    return launch_status();
}

template <typename R, int N, int P, int K>
int run_loss_backward(const kvae_dims* d, const kvae_loss_bwd* io, void* ws, size_t ws_bytes,
                      hipStream_t s) {
    using PG = LossParamGrads<N, P, K>;
    if (loss_form(d) == 16) return row_loss_backward(d, io, ws, ws_bytes, s, 0);
    const long rows = loss_rows(d);
    const size_t bars = (size_t)loss_bars_elems(d->T, d->B, d->n) * sizeof(R);
    if (ws_bytes < bars + (size_t)rows * PG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
    LossBwdArgs<R> g;
    g.f = make_loss_args<R>(d, &io->fwd, nullptr);
    g.burn_in = d->burn_in;
    g.g_scale = (const R*)io->g_scale;
    g.g_scale_stride = io->g_scale_stride;
    g.w_obs = (R)io->fwd.w_obs;
    g.w_trans = (R)io->fwd.w_trans;
    g.w_post = (R)io->fwd.w_post;
    g.d_a = (R*)io->d_a;
    g.d_alpha = (R*)io->d_alpha;
    loss_bars_pointers<R>(ws, d->T, d->B, d->n, g);
    g.partial = (R*)((char*)ws + bars);
    bool launched = false;
    if constexpr (N <= 4 && P <= 4) {
        if (loss_use_quad(d)) {
            hipLaunchKernelGGL((loss_backward_quad_kernel<R, N, P, K>), dim3(rows),
                               dim3(LOSS_BLOCK), 0, s, g);
            launched = true;
        }
    }
    if (!launched)
        hipLaunchKernelGGL((loss_backward_kernel<R, N, P, K>), dim3(rows), dim3(LOSS_BLOCK), 0, s,
                           g);
    ReduceSegments seg;
    void* outs[6] = {io->d_A_K, io->d_C_K, io->d_Q, io->d_R, io->d_m0, io->d_P0};
    const int begin[7] = {PG::OFF_A, PG::OFF_C, PG::OFF_Q, PG::OFF_R, PG::OFF_M0, PG::OFF_P0, PG::LEN};
    seg.count = 6;
    for (int i = 0; i < 6; ++i) seg.out[i] = outs[i];
    for (int i = 0; i < 7; ++i) seg.begin[i] = begin[i];
    hipLaunchKernelGGL((reduce_rows_kernel<R>), dim3(PG::LEN), dim3(256), 0, s,
                       (const R*)g.partial, rows, (int)PG::LEN, seg);
    return launch_status();
}

static int loss_dims_supported(int n, int p, int K) {
#define KVAE_SUP(n_, p_, K_) \
    if (n == n_ && p == p_ && K == K_) return 1;
    KVAE_DIM_LIST(KVAE_SUP)
#undef KVAE_SUP
    return 0;
}

size_t loss_workspace_bytes(const kvae_dims* d, int op) {
    if (!loss_dims_supported(d->n, d->p, d->K)) return 0;
    const size_t el = d->dtype == KVAE_F32 ? 4 : 8;
    const long n = d->n, p = d->p, K = d->K;
    const long rows = loss_rows_max(d->n, d->p, d->B);
    // the row form (float, n and p up to 4) keeps one row of sums per four sequences
    const size_t row_form = (d->n <= 4 && d->p <= 4) ? row_loss_workspace_bytes(d) : 0;
    const size_t fwd = (size_t)rows * 3 * el;
    const size_t bwd = ((size_t)loss_bars_elems(d->T, d->B, n) +
                        (size_t)rows * (K * n * n + K * p * n + n * n + p * p + n + n * n)) * el;
    if (op == KVAE_OP_LOSS_FWD) return fwd > row_form ? fwd : row_form;
    if (op == KVAE_OP_LOSS_BWD || op == KVAE_OP_LOSS_VALUE_AND_GRAD)
        return bwd > row_form ? bwd : row_form;
    return 0;
}

}  // namespace kvae

using namespace kvae;

#define KVAE_DISPATCH_CASE(n_, p_, K_)                                  \
    if (d->n == n_ && d->p == p_ && d->K == K_) {                       \
        if (d->dtype == KVAE_F32) return KVAE_CALL(float, n_, p_, K_);  \
        return KVAE_CALL(double, n_, p_, K_);                           \
    }

extern "C" {

int kvae_loss_supported(int n, int p, int K, int dtype) {
    if (dtype != KVAE_F32 && dtype != KVAE_F64) return 0;
    return loss_dims_supported(n, p, K);
}

size_t kvae_loss_saved_bytes(const kvae_dims* d) {
    if (!dims_ok(d)) return 0;
    if (loss_form(d) == 16) return 256;   // the row form saves nothing: its backward recomputes
    return (size_t)loss_saved_elems(d->T, d->B, d->n) * (d->dtype == KVAE_F32 ? 4 : 8);
}

int kvae_kalman_loss_forward(const kvae_dims* d, const kvae_loss_fwd* io, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !io->a || !io->alpha || !io->A_K || !io->C_K || !io->Q || !io->R ||
        !io->m0 || !io->P0 || !io->eps || !io->saved || !io->out || !workspace)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_loss_forward<R, n_, p_, K_>(d, io, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_kalman_loss_backward(const kvae_dims* d, const kvae_loss_bwd* io, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !io->fwd.a || !io->fwd.alpha || !io->fwd.A_K || !io->fwd.C_K ||
        !io->fwd.Q || !io->fwd.R || !io->fwd.m0 || !io->fwd.P0 || !io->fwd.eps ||
        !io->fwd.saved || !io->g_scale || !io->d_a || !io->d_alpha || !workspace)
        return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_loss_backward<R, n_, p_, K_>(d, io, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

int kvae_kalman_loss_value_and_grad(const kvae_dims* d, const kvae_loss_bwd* io, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !io->fwd.a || !io->fwd.alpha || !io->fwd.A_K || !io->fwd.C_K ||
        !io->fwd.Q || !io->fwd.R || !io->fwd.m0 || !io->fwd.P0 || !io->fwd.eps || !io->fwd.out ||
        !io->d_a || !io->d_alpha || !workspace)
        return KVAE_E_BADARG;
    if (!loss_dims_supported(d->n, d->p, d->K)) return KVAE_E_UNSUPPORTED;
    if (loss_form(d) == 16)
        return row_loss_backward(d, io, workspace, workspace_bytes, (hipStream_t)stream, 1);
    // the other forms hand their state over through `saved`: two launches
    if (!io->fwd.saved || !io->g_scale) return KVAE_E_BADARG;
    const int rc = kvae_kalman_loss_forward(d, &io->fwd, workspace, workspace_bytes, stream);
    if (rc != KVAE_OK) return rc;
    return kvae_kalman_loss_backward(d, io, workspace, workspace_bytes, stream);
}

}  // extern "C"
