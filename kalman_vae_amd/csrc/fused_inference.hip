// kvae_filter_smooth_fused: filter (optionally with the perceptron weight network in the
// loop) and RTS smoother of every sequence in one launch, one lane per sequence
// (kalman_filter_smooth.h).  Outputs the caller does not ask for are not written; the
// filtered moments the reverse sweep reads back go through the workspace when they are not
// outputs.
#include "launch_common.h"
#include "kalman_filter_smooth.h"

namespace kvae {

constexpr int FS_BLOCK = 64;   // one wavefront per workgroup
constexpr int FS_NET_MAX_BYTES = 64 * 1024;

template <typename R, int N, int P, int K, bool NET>
__global__ void __launch_bounds__(FS_BLOCK)
filter_smooth_kernel(FilterSmoothArgs<R> g, MaskedNetArgs<R> q, const R* __restrict__ net) {
    // `net` is read-only and not aliased: its (uniform) loads are scalar loads
    q.net = net;
    const long b = (long)blockIdx.x * FS_BLOCK + threadIdx.x;
    if (b < g.f.B) filter_smooth_sequence<R, N, P, K>(g, NET ? &q : nullptr, (int)b);
}

template <typename R>
size_t filter_smooth_workspace_elems(const kvae_dims* d, const kvae_filter_fwd* f) {
    const size_t tb = (size_t)d->T * d->B, n = d->n;
    size_t e = 0;
    if (!f || !f->mean_filt) e += tb * n;
    if (!f || !f->cov_filt) e += tb * n * n;
    return e;
}

template <typename R, int N, int P, int K>
int run_filter_smooth(const kvae_dims* d, const kvae_filter_fwd* f, const kvae_masked_mlp* net,
                      const kvae_smooth_fwd* s, void* ws, size_t ws_bytes, hipStream_t st) {
    if (net && (size_t)(net->hidden * (P + 1 + K) + K) * sizeof(R) > (size_t)FS_NET_MAX_BYTES)
        return KVAE_E_UNSUPPORTED;
    if (ws_bytes < filter_smooth_workspace_elems<R>(d, f) * sizeof(R)) return KVAE_E_WORKSPACE;
    FilterSmoothArgs<R> g;
    g.f.T = d->T; g.f.B = d->B; g.f.symmetrize = d->symmetrize;
    g.f.carry_per_sequence = f->carry_per_sequence;
    g.f.a = (const R*)f->a; g.f.alpha = (const R*)f->alpha; g.f.mask = (const R*)f->mask;
    g.f.A_K = (const R*)f->A_K; g.f.C_K = (const R*)f->C_K; g.f.Q = (const R*)f->Q;
    g.f.Rm = (const R*)f->R; g.f.m_in = (const R*)f->m_in; g.f.P_in = (const R*)f->P_in;
    R* w = (R*)ws;
    const size_t tb = (size_t)d->T * d->B;
    auto out_or_ws = [&](void* out, size_t elems) {
        if (out) return (R*)out;
        R* p = w;
        w += elems;
        return p;
    };
    g.f.m = out_or_ws(f->mean_filt, tb * N);
    g.f.Pf = out_or_ws(f->cov_filt, tb * N * N);
    g.f.mn = (R*)f->mean_pred;     // written only when asked for: the reverse sweep
    g.f.Pn = (R*)f->cov_pred;      // recomputes the predictions
    g.f.As = (R*)f->mat_As; g.f.Cs = (R*)f->mat_Cs; g.f.status = f->status;
    g.LR = (const R*)s->LR; g.eps_s = (const R*)s->eps_state; g.eps_o = (const R*)s->eps_obs;
    g.ms = (R*)s->mean_smooth; g.Ps = (R*)s->cov_smooth; g.zs = (R*)s->zs;
    g.ares = (R*)s->as_resampled;
    MaskedNetArgs<R> q{};
    const dim3 grid((unsigned)((d->B + FS_BLOCK - 1) / FS_BLOCK)), block(FS_BLOCK);
    if (net) {
        q = MaskedNetArgs<R>{net->hidden,        nullptr,              (const R*)net->eps_z,
                             (const R*)net->L_Q, (const R*)net->eps_a, (const R*)net->L_R,
                             (R*)net->a_w,       (R*)f->alpha};
        hipLaunchKernelGGL((filter_smooth_kernel<R, N, P, K, true>), grid, block, 0, st, g, q,
                           (const R*)net->packed);
    } else {
        hipLaunchKernelGGL((filter_smooth_kernel<R, N, P, K, false>), grid, block, 0, st, g, q,
                           (const R*)nullptr);
    }
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

size_t kvae_filter_smooth_fused_workspace_bytes(const kvae_dims* d, const kvae_filter_fwd* f) {
    if (!dims_ok(d)) return 0;
    return filter_smooth_workspace_elems<char>(d, f) * (d->dtype == KVAE_F32 ? 4 : 8);
}

int kvae_filter_smooth_fused(const kvae_dims* d, const kvae_filter_fwd* f,
                             const kvae_masked_mlp* net, const kvae_smooth_fwd* s,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!dims_ok(d) || !f || !s || !f->a || !f->alpha || !f->A_K || !f->C_K || !f->Q || !f->R ||
        !f->m_in || !f->P_in)
        return KVAE_E_BADARG;
    if (net && (!net->packed || net->hidden <= 0 || !net->a_w)) return KVAE_E_BADARG;
    if (net && ((net->eps_z && !net->L_Q) || (net->eps_a && !net->L_R))) return KVAE_E_BADARG;
    if (s->eps_obs && !s->LR) return KVAE_E_BADARG;
    if (kvae_filter_smooth_fused_workspace_bytes(d, f) > 0 && !workspace) return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) \
    run_filter_smooth<R, n_, p_, K_>(d, f, net, s, workspace, workspace_bytes, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}

}  // extern "C"
