// kvae_filter_backward_masked_mlp: adjoint of kvae_filter_forward_masked_mlp, the masked
// filter with the perceptron weight network evaluated inside the recursion
// (kvae/state_space_model.py:219-352 with an observation mask).  One lane per sequence,
// reverse in time: the filter adjoint of kalman_core.h with, per step, the gradient of
// the weights sent on through the softmax, the perceptron, the blend and the imputation
// (MlpNetAdjoint).  Parameter gradients are summed per wavefront and reduced by a second,
// deterministic launch.
#include "launch_common.h"

namespace kvae {

constexpr int MB_BLOCK = 64;   // one wavefront per workgroup
constexpr int MB_NET_MAX_BYTES = 64 * 1024;

template <typename R, int N, int P, int K>
struct MaskedParamGrads {
    using PG = FilterParamGrads<N, P, K>;
    static constexpr int OFF_LQ = PG::LEN;
    static constexpr int OFF_LR = OFF_LQ + N * N;
    static constexpr int LEN = OFF_LR + P * P;
};

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(MB_BLOCK)
filter_backward_masked_mlp_kernel(FilterBwdArgs<R> g, MlpNetAdjoint<R> net,
                                  const R* __restrict__ packed) {
    using MG = MaskedParamGrads<R, N, P, K>;
    // read-only, non-aliased: the (uniform) loads of the network are scalar loads
    net.q.net = packed;
    const long b = (long)blockIdx.x * MB_BLOCK + threadIdx.x;
    R acc[MG::LEN];
    vzero<R, MG::LEN>(acc);
    net.acc_LQ = acc + MG::OFF_LQ;
    net.acc_LR = acc + MG::OFF_LR;
    if (b < g.f.B) filter_backward_masked_mlp_sequence<R, N, P, K>(g, net, (int)b, acc);
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int e = 0; e < MG::LEN; ++e) {
        const R s = wave_sum<R>(acc[e]);
        if (lane == 0) g.partial[(long)blockIdx.x * MG::LEN + e] = s;
    }
}

template <typename R, int N, int P, int K>
int run_masked_backward(const kvae_dims* d, const kvae_filter_bwd* io,
                        const kvae_masked_mlp_bwd* nb, void* ws, size_t ws_bytes, hipStream_t s) {
    using MG = MaskedParamGrads<R, N, P, K>;
    using PG = FilterParamGrads<N, P, K>;
    if ((size_t)(nb->fwd.hidden * (P + 1 + K) + K) * sizeof(R) > (size_t)MB_NET_MAX_BYTES)
        return KVAE_E_UNSUPPORTED;
    const long rows = (d->B + MB_BLOCK - 1) / MB_BLOCK;
    if (ws_bytes < (size_t)rows * MG::LEN * sizeof(R)) return KVAE_E_WORKSPACE;
    FilterBwdArgs<R> g = make_args<R>(d, io, ws);
    MlpNetAdjoint<R> net;
    net.q = MaskedNetArgs<R>{nb->fwd.hidden,        nullptr,                  (const R*)nb->fwd.eps_z,
                             (const R*)nb->fwd.L_Q, (const R*)nb->fwd.eps_a, (const R*)nb->fwd.L_R,
                             (R*)nb->fwd.a_w,       nullptr};
    net.g_a_w = (const R*)nb->g_a_w;
    net.g_alpha = (const R*)nb->g_alpha;
    net.d_logits = (R*)nb->d_logits;
    net.learn = nb->learn_weight_model;
    net.acc_LQ = net.acc_LR = nullptr;
    hipLaunchKernelGGL((filter_backward_masked_mlp_kernel<R, N, P, K>), dim3((unsigned)rows),
                       dim3(MB_BLOCK), 0, s, g, net, (const R*)nb->fwd.packed);
    void* outs[6] = {io->d_A_K, io->d_C_K, io->d_Q, io->d_R, nb->d_L_Q, nb->d_L_R};
    const int begin[7] = {PG::OFF_A, PG::OFF_C, PG::OFF_Q, PG::OFF_R, MG::OFF_LQ, MG::OFF_LR, MG::LEN};
    launch_reduce<R>(ws, rows, MG::LEN, 6, outs, begin, s);
    return launch_status();
}

}  // namespace kvae

using namespace kvae;

#define KVAE_DISPATCH_CASE(n_, p_, K_)                                  \
    if (d->n == n_ && d->p == p_ && d->K == K_) {                       \
        if (d->dtype == KVAE_F32) return KVAE_CALL(float, n_, p_, K_);  \
        return KVAE_CALL(double, n_, p_, K_);                           \
    }

extern "C" {

size_t kvae_filter_backward_masked_mlp_workspace_bytes(const kvae_dims* d) {
    if (!dims_ok(d)) return 0;
    const size_t n = d->n, p = d->p, K = d->K;
    const size_t len = K * n * n + K * p * n + n * n + p * p + n * n + p * p;
    return (size_t)((d->B + MB_BLOCK - 1) / MB_BLOCK) * len * (d->dtype == KVAE_F32 ? 4 : 8);
}

int kvae_filter_backward_masked_mlp(const kvae_dims* d, const kvae_filter_bwd* io,
                                    const kvae_masked_mlp_bwd* net, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !io || !net || !io->fwd.a || !io->fwd.alpha || !io->fwd.A_K ||
        !io->fwd.C_K || !io->fwd.Q || !io->fwd.R || !io->fwd.m_in || !io->fwd.P_in ||
        !io->fwd.mean_filt || !io->fwd.cov_filt || !io->fwd.mean_pred || !io->fwd.cov_pred ||
        !io->d_a || !io->d_alpha || !io->d_m_in || !io->d_P_in || !workspace)
        return KVAE_E_BADARG;
    if (!net->fwd.packed || net->fwd.hidden <= 0 || !net->fwd.a_w || !net->d_logits)
        return KVAE_E_BADARG;
    if ((net->fwd.eps_z && !net->fwd.L_Q) || (net->fwd.eps_a && !net->fwd.L_R)) return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_masked_backward<R, n_, p_, K_>(d, io, net, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

}  // extern "C"
