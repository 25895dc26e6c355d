// TEST-ONLY host build of the per-sequence device math.
//
// kalman_core.h is written so that the code a GPU lane executes for one
// sequence is an ordinary C++ function.  This file compiles that same source
// with g++ and runs it in a loop over the batch on the CPU, behind functions
// that take the structs of include/kvae_kalman.h.  It exists so that the math
// and the adjoints can be checked against the oracle in a container without a
// GPU.  It is not part of the package, is never loaded by it, and is not a
// fallback: kalman_vae_amd refuses to run without its HIP library.
#include <string.h>

#include <vector>

#include "../../include/kvae_kalman.h"
#include "../../kalman_vae_amd/csrc/kalman_core.h"
#include "../../kalman_vae_amd/csrc/kalman_loss.h"
#include "../../kalman_vae_amd/csrc/kalman_rollout.h"
#include "../../kalman_vae_amd/csrc/kalman_filter_smooth.h"
#include "../../kalman_vae_amd/csrc/kalman_wave.h"
#include "../../kalman_vae_amd/csrc/kalman_row.h"
#include <memory>

using namespace kvae;

namespace {

template <typename R>
FilterFwdArgs<R> mk(const kvae_dims* d, const kvae_filter_fwd* io) {
    FilterFwdArgs<R> g;
    g.T = d->T; g.B = d->B; g.symmetrize = d->symmetrize;
    g.carry_per_sequence = io->carry_per_sequence;
    g.a = (const R*)io->a; g.alpha = (const R*)io->alpha; g.mask = (const R*)io->mask;
    g.A_K = (const R*)io->A_K; g.C_K = (const R*)io->C_K; g.Q = (const R*)io->Q;
    g.Rm = (const R*)io->R; g.m_in = (const R*)io->m_in; g.P_in = (const R*)io->P_in;
    g.m = (R*)io->mean_filt; g.Pf = (R*)io->cov_filt; g.mn = (R*)io->mean_pred;
    g.Pn = (R*)io->cov_pred; g.As = (R*)io->mat_As; g.Cs = (R*)io->mat_Cs;
    g.status = io->status;
    return g;
}
template <typename R>
SmoothFwdArgs<R> mk(const kvae_dims* d, const kvae_smooth_fwd* io) {
    SmoothFwdArgs<R> g;
    g.T = d->T; g.B = d->B; g.symmetrize = d->symmetrize;
    g.m = (const R*)io->mean_filt; g.Pf = (const R*)io->cov_filt;
    g.mn = (const R*)io->mean_pred; g.Pn = (const R*)io->cov_pred;
    g.As = (const R*)io->mat_As; g.Cs = (const R*)io->mat_Cs; g.LR = (const R*)io->LR;
    g.eps_s = (const R*)io->eps_state; g.eps_o = (const R*)io->eps_obs;
    g.ms = (R*)io->mean_smooth; g.Ps = (R*)io->cov_smooth; g.zs = (R*)io->zs;
    g.ares = (R*)io->as_resampled; g.status = io->status;
    return g;
}
template <typename R>
ElboFwdArgs<R> mk(const kvae_dims* d, const kvae_elbo_fwd* io) {
    ElboFwdArgs<R> g;
    g.T = d->T; g.B = d->B;
    g.a = (const R*)io->a; g.ms = (const R*)io->mean_smooth; g.Ps = (const R*)io->cov_smooth;
    g.As = (const R*)io->mat_As; g.Cs = (const R*)io->mat_Cs; g.Q = (const R*)io->Q;
    g.Rm = (const R*)io->R; g.m0 = (const R*)io->m0; g.P0 = (const R*)io->P0;
    g.eps = (const R*)io->eps; g.obs = (R*)io->obs; g.trans = (R*)io->trans;
    g.post = (R*)io->post; g.z = (R*)io->z; g.status = io->status;
    return g;
}

template <typename R>
void put(void* out, const R* src, int len) {
    if (out) memcpy(out, src, sizeof(R) * len);
}

template <typename R, int N, int P, int K>
int filter_fwd(const kvae_dims* d, const kvae_filter_fwd* io) {
    FilterFwdArgs<R> g = mk<R>(d, io);
    for (int b = 0; b < d->B; ++b) filter_forward_sequence<R, N, P, K>(g, b);
    return 0;
}
template <typename R, int N, int P, int K>
int masked_mlp_fwd(const kvae_dims* d, const kvae_filter_fwd* io, const kvae_masked_mlp* net) {
    FilterFwdArgs<R> g = mk<R>(d, io);
    MaskedNetArgs<R> q{net->hidden,        (const R*)net->packed, (const R*)net->eps_z,
                       (const R*)net->L_Q, (const R*)net->eps_a,  (const R*)net->L_R,
                       (R*)net->a_w,       (R*)io->alpha};
    for (int b = 0; b < d->B; ++b) filter_forward_masked_mlp_sequence<R, N, P, K>(g, q, b);
    return 0;
}
template <typename R, int N, int P, int K>
int masked_mlp_bwd(const kvae_dims* d, const kvae_filter_bwd* io, const kvae_masked_mlp_bwd* nb) {
    using PG = FilterParamGrads<N, P, K>;
    constexpr int LEN = PG::LEN + N * N + P * P;
    FilterBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_m = (const R*)io->g_mean_filt; g.g_P = (const R*)io->g_cov_filt;
    g.g_mn = (const R*)io->g_mean_pred; g.g_Pn = (const R*)io->g_cov_pred;
    g.g_As = (const R*)io->g_mat_As; g.g_Cs = (const R*)io->g_mat_Cs;
    g.d_a = (R*)io->d_a; g.d_alpha = (R*)io->d_alpha;
    g.d_m_in = (R*)io->d_m_in; g.d_P_in = (R*)io->d_P_in; g.partial = nullptr;
    g.bar_As = nullptr; g.bar_Cs = nullptr;
    std::vector<R> acc(LEN, R(0));
    MlpNetAdjoint<R> net;
    net.q = MaskedNetArgs<R>{nb->fwd.hidden,        (const R*)nb->fwd.packed, (const R*)nb->fwd.eps_z,
                             (const R*)nb->fwd.L_Q, (const R*)nb->fwd.eps_a,  (const R*)nb->fwd.L_R,
                             (R*)nb->fwd.a_w,       nullptr};
    net.g_a_w = (const R*)nb->g_a_w; net.g_alpha = (const R*)nb->g_alpha;
    net.d_logits = (R*)nb->d_logits; net.learn = nb->learn_weight_model;
    net.acc_LQ = acc.data() + PG::LEN; net.acc_LR = acc.data() + PG::LEN + N * N;
    for (int b = 0; b < d->B; ++b)
        filter_backward_masked_mlp_sequence<R, N, P, K>(g, net, b, acc.data());
    put<R>(io->d_A_K, acc.data() + PG::OFF_A, K * N * N);
    put<R>(io->d_C_K, acc.data() + PG::OFF_C, K * P * N);
    put<R>(io->d_Q, acc.data() + PG::OFF_Q, N * N);
    put<R>(io->d_R, acc.data() + PG::OFF_R, P * P);
    put<R>(nb->d_L_Q, acc.data() + PG::LEN, N * N);
    put<R>(nb->d_L_R, acc.data() + PG::LEN + N * N, P * P);
    return 0;
}
// every output present (the caller allocates all of them): checks the fused statements
template <typename R, int N, int P, int K>
int filter_smooth(const kvae_dims* d, const kvae_filter_fwd* f, const kvae_masked_mlp* net,
                  const kvae_smooth_fwd* s) {
    FilterSmoothArgs<R> g;
    g.f = mk<R>(d, f);
    g.LR = (const R*)s->LR; g.eps_s = (const R*)s->eps_state; g.eps_o = (const R*)s->eps_obs;
    g.ms = (R*)s->mean_smooth; g.Ps = (R*)s->cov_smooth; g.zs = (R*)s->zs;
    g.ares = (R*)s->as_resampled;
    MaskedNetArgs<R> q{};
    if (net)
        q = MaskedNetArgs<R>{net->hidden,        (const R*)net->packed, (const R*)net->eps_z,
                             (const R*)net->L_Q, (const R*)net->eps_a,  (const R*)net->L_R,
                             (R*)net->a_w,       (R*)f->alpha};
    for (int b = 0; b < d->B; ++b) filter_smooth_sequence<R, N, P, K>(g, net ? &q : nullptr, b);
    return 0;
}
template <typename R, int N, int P, int K>
int rollout(const kvae_dims* d, const kvae_rollout* io) {
    RolloutArgs<R> g;
    g.S = d->T; g.B = d->B; g.hidden = io->hidden; g.phase = io->phase; g.sample = io->sample;
    g.A_K = (const R*)io->A_K; g.C_K = (const R*)io->C_K; g.Q = (const R*)io->Q;
    g.mn_in = (const R*)io->mean_pred; g.Pn_in = (const R*)io->cov_pred;
    g.C_in = (const R*)io->mat_C; g.eps = (const R*)io->eps; g.net = (const R*)io->packed_mlp;
    g.alpha = (const R*)io->alpha;
    g.as_new = (R*)io->as_new; g.means = (R*)io->means; g.covs = (R*)io->covs;
    g.mn_new = (R*)io->mean_pred_new; g.Pn_new = (R*)io->cov_pred_new;
    g.As_new = (R*)io->mat_As_new; g.Cs_new = (R*)io->mat_Cs_new; g.status = io->status;
    for (int b = 0; b < d->B; ++b) rollout_sequence<R, N, P, K>(g, b);
    return 0;
}
template <typename R, int N, int P, int K>
int filter_bwd(const kvae_dims* d, const kvae_filter_bwd* io) {
    using PG = FilterParamGrads<N, P, K>;
    FilterBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_m = (const R*)io->g_mean_filt; g.g_P = (const R*)io->g_cov_filt;
    g.g_mn = (const R*)io->g_mean_pred; g.g_Pn = (const R*)io->g_cov_pred;
    g.g_As = (const R*)io->g_mat_As; g.g_Cs = (const R*)io->g_mat_Cs;
    g.d_a = (R*)io->d_a; g.d_alpha = (R*)io->d_alpha;
    g.d_m_in = (R*)io->d_m_in; g.d_P_in = (R*)io->d_P_in; g.partial = nullptr;
    g.bar_As = nullptr; g.bar_Cs = nullptr;
    std::vector<R> acc(PG::LEN, R(0));
    for (int b = 0; b < d->B; ++b) filter_backward_sequence<R, N, P, K>(g, b, acc.data());
    put<R>(io->d_A_K, acc.data() + PG::OFF_A, K * N * N);
    put<R>(io->d_C_K, acc.data() + PG::OFF_C, K * P * N);
    put<R>(io->d_Q, acc.data() + PG::OFF_Q, N * N);
    put<R>(io->d_R, acc.data() + PG::OFF_R, P * P);
    return 0;
}
template <typename R, int N, int P, int K>
int smooth_fwd(const kvae_dims* d, const kvae_smooth_fwd* io) {
    SmoothFwdArgs<R> g = mk<R>(d, io);
    for (int b = 0; b < d->B; ++b) smooth_forward_sequence<R, N, P>(g, b);
    return 0;
}
template <typename R, int N, int P, int K>
int smooth_bwd(const kvae_dims* d, const kvae_smooth_bwd* io) {
    SmoothBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_ms = (const R*)io->g_mean_smooth; g.g_Ps = (const R*)io->g_cov_smooth;
    g.g_zs = (const R*)io->g_zs; g.g_ares = (const R*)io->g_as_resampled;
    g.d_m = (R*)io->d_mean_filt; g.d_P = (R*)io->d_cov_filt; g.d_mn = (R*)io->d_mean_pred;
    g.d_Pn = (R*)io->d_cov_pred; g.d_As = (R*)io->d_mat_As; g.d_Cs = (R*)io->d_mat_Cs;
    g.partial = nullptr;
    std::vector<R> acc(P * P, R(0));
    for (int b = 0; b < d->B; ++b) smooth_backward_sequence<R, N, P>(g, b, acc.data());
    put<R>(io->d_LR, acc.data(), P * P);
    return 0;
}
template <typename R, int N, int P, int K>
int elbo_fwd(const kvae_dims* d, const kvae_elbo_fwd* io) {
    ElboFwdArgs<R> g = mk<R>(d, io);
    for (int t = 0; t < d->T; ++t)
        for (int b = 0; b < d->B; ++b) elbo_forward_pair<R, N, P>(g, t, b);
    return 0;
}
template <typename R, int N, int P, int K>
int elbo_bwd(const kvae_dims* d, const kvae_elbo_bwd* io) {
    using PG = ElboParamGrads<N, P>;
    ElboBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.g_obs = (const R*)io->g_obs; g.g_trans = (const R*)io->g_trans;
    g.g_post = (const R*)io->g_post;
    g.g_scale = (const R*)io->g_scale; g.g_scale_stride = io->g_scale_stride;
    g.w_obs = (R)io->w_obs; g.w_trans = (R)io->w_trans; g.w_post = (R)io->w_post;
    g.d_a = (R*)io->d_a; g.d_ms = (R*)io->d_mean_smooth; g.d_Ps = (R*)io->d_cov_smooth;
    g.d_As = (R*)io->d_mat_As; g.d_Cs = (R*)io->d_mat_Cs; g.partial = nullptr;
    std::vector<R> acc(PG::LEN, R(0));
    for (int t = 0; t < d->T; ++t)
        for (int b = 0; b < d->B; ++b) elbo_backward_pair<R, N, P>(g, t, b, acc.data());
    put<R>(io->d_Q, acc.data() + PG::OFF_Q, N * N);
    put<R>(io->d_R, acc.data() + PG::OFF_R, P * P);
    put<R>(io->d_m0, acc.data() + PG::OFF_M0, N);
    put<R>(io->d_P0, acc.data() + PG::OFF_P0, N * N);
    return 0;
}

template <typename R>
LossFwdArgs<R> mkl(const kvae_dims* d, const kvae_loss_fwd* io) {
    LossFwdArgs<R> g;
    g.T = d->T; g.B = d->B; g.symmetrize = d->symmetrize;
    g.a = (const R*)io->a; g.alpha = (const R*)io->alpha; g.mask = (const R*)io->mask;
    g.A_K = (const R*)io->A_K; g.C_K = (const R*)io->C_K; g.Q = (const R*)io->Q;
    g.Rm = (const R*)io->R; g.m0 = (const R*)io->m0; g.P0 = (const R*)io->P0;
    g.eps = (const R*)io->eps;
    loss_saved_pointers<R>(io->saved, d->T, d->B, d->n, g);
    g.terms = (R*)io->terms; g.partial = nullptr; g.status = io->status;
    return g;
}
template <typename R, int N, int P, int K>
int loss_fwd(const kvae_dims* d, const kvae_loss_fwd* io) {
    LossFwdArgs<R> g = mkl<R>(d, io);
    R sums[3] = {R(0), R(0), R(0)};
    for (int b = 0; b < d->B; ++b) loss_forward_sequence<R, N, P, K>(g, b, sums);
    R* out = (R*)io->out;
    out[0] = (R)io->w_obs * sums[0];
    out[1] = (R)io->w_trans * sums[1];
    out[2] = (R)io->w_post * sums[2];
    out[3] = out[0] + out[1] + out[2];
    return 0;
}
template <typename R, int N, int P, int K>
int loss_bwd(const kvae_dims* d, const kvae_loss_bwd* io) {
    using PG = LossParamGrads<N, P, K>;
    LossBwdArgs<R> g;
    g.f = mkl<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_scale = (const R*)io->g_scale; g.g_scale_stride = io->g_scale_stride;
    g.w_obs = (R)io->fwd.w_obs; g.w_trans = (R)io->fwd.w_trans; g.w_post = (R)io->fwd.w_post;
    g.d_a = (R*)io->d_a; g.d_alpha = (R*)io->d_alpha; g.partial = nullptr;
    std::vector<R> bars(loss_bars_elems(d->T, d->B, d->n));
    loss_bars_pointers<R>(bars.data(), d->T, d->B, d->n, g);
    std::vector<R> acc(PG::LEN, R(0));
    for (int b = 0; b < d->B; ++b) loss_backward_sequence<R, N, P, K>(g, b, acc.data());
    put<R>(io->d_A_K, acc.data() + PG::OFF_A, K * N * N);
    put<R>(io->d_C_K, acc.data() + PG::OFF_C, K * P * N);
    put<R>(io->d_Q, acc.data() + PG::OFF_Q, N * N);
    put<R>(io->d_R, acc.data() + PG::OFF_R, P * P);
    put<R>(io->d_m0, acc.data() + PG::OFF_M0, N);
    put<R>(io->d_P0, acc.data() + PG::OFF_P0, N * N);
    return 0;
}

}  // namespace


// ---- wavefront-per-sequence source (kalman_wave.h) run by one host thread ----
namespace {
template <typename R>
FilterBwdArgs<R> mkb(const kvae_dims* d, const kvae_filter_bwd* io) {
    FilterBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_m = (const R*)io->g_mean_filt; g.g_P = (const R*)io->g_cov_filt;
    g.g_mn = (const R*)io->g_mean_pred; g.g_Pn = (const R*)io->g_cov_pred;
    g.g_As = (const R*)io->g_mat_As; g.g_Cs = (const R*)io->g_mat_Cs;
    g.d_a = (R*)io->d_a; g.d_alpha = (R*)io->d_alpha;
    g.d_m_in = (R*)io->d_m_in; g.d_P_in = (R*)io->d_P_in; g.partial = nullptr;
    g.bar_As = nullptr; g.bar_Cs = nullptr;
    return g;
}
template <typename R>
SmoothBwdArgs<R> mkb(const kvae_dims* d, const kvae_smooth_bwd* io) {
    SmoothBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.burn_in = d->burn_in;
    g.g_ms = (const R*)io->g_mean_smooth; g.g_Ps = (const R*)io->g_cov_smooth;
    g.g_zs = (const R*)io->g_zs; g.g_ares = (const R*)io->g_as_resampled;
    g.d_m = (R*)io->d_mean_filt; g.d_P = (R*)io->d_cov_filt; g.d_mn = (R*)io->d_mean_pred;
    g.d_Pn = (R*)io->d_cov_pred; g.d_As = (R*)io->d_mat_As; g.d_Cs = (R*)io->d_mat_Cs;
    g.partial = nullptr;
    return g;
}
template <typename R>
ElboBwdArgs<R> mkb(const kvae_dims* d, const kvae_elbo_bwd* io) {
    ElboBwdArgs<R> g;
    g.f = mk<R>(d, &io->fwd);
    g.g_obs = (const R*)io->g_obs; g.g_trans = (const R*)io->g_trans;
    g.g_post = (const R*)io->g_post;
    g.g_scale = (const R*)io->g_scale; g.g_scale_stride = io->g_scale_stride;
    g.w_obs = (R)io->w_obs; g.w_trans = (R)io->w_trans; g.w_post = (R)io->w_post;
    g.d_a = (R*)io->d_a; g.d_ms = (R*)io->d_mean_smooth; g.d_Ps = (R*)io->d_cov_smooth;
    g.d_As = (R*)io->d_mat_As; g.d_Cs = (R*)io->d_mat_Cs; g.partial = nullptr;
    return g;
}

template <typename R, int N, int P, int K>
int wfilter_fwd(const kvae_dims* d, const kvae_filter_fwd* io) {
    FilterFwdArgs<R> g = mk<R>(d, io);
    auto s = std::make_unique<wave::FilterScratch<R, N, P, K>>();
    for (int b = 0; b < d->B; ++b) wave::filter_forward_sequence<R, N, P, K>(g, b, *s);
    return 0;
}
template <typename R, int N, int P, int K>
int wfilter_bwd(const kvae_dims* d, const kvae_filter_bwd* io) {
    FilterBwdArgs<R> g = mkb<R>(d, io);
    const long rows = (long)(d->T + 1) * d->B;
    std::vector<R> bar_As(rows * N * N), bar_Cs(rows * P * N);
    g.bar_As = bar_As.data();
    g.bar_Cs = bar_Cs.data();
    auto s = std::make_unique<wave::FilterBwdScratch<R, N, P, K>>();
    auto acc = std::make_unique<wave::FilterGradAcc<R, N, P, K>>();
    acc->zero();
    for (int b = 0; b < d->B; ++b) wave::filter_backward_sequence<R, N, P, K>(g, b, *s, *acc);
    std::vector<R> row(N * N + P * P);
    acc->store(row.data());
    put<R>(io->d_Q, row.data(), N * N);
    put<R>(io->d_R, row.data() + N * N, P * P);
    std::vector<R> dA(K * N * N), dC(K * P * N);
    wave::mix_adjoint_host<R, N, P, K>(rows, g.f.alpha, g.bar_As, g.bar_Cs, g.f.A_K, g.f.C_K,
                                       g.d_alpha, dA.data(), dC.data());
    put<R>(io->d_A_K, dA.data(), K * N * N);
    put<R>(io->d_C_K, dC.data(), K * P * N);
    return 0;
}
template <typename R, int N, int P, int K>
int wsmooth_fwd(const kvae_dims* d, const kvae_smooth_fwd* io) {
    SmoothFwdArgs<R> g = mk<R>(d, io);
    auto s = std::make_unique<wave::SmoothScratch<R, N, P>>();
    for (int b = 0; b < d->B; ++b) wave::smooth_forward_sequence<R, N, P>(g, b, *s);
    return 0;
}
template <typename R, int N, int P, int K>
int wsmooth_bwd(const kvae_dims* d, const kvae_smooth_bwd* io) {
    SmoothBwdArgs<R> g = mkb<R>(d, io);
    auto s = std::make_unique<wave::SmoothBwdScratch<R, N, P>>();
    wave::Acc<R, P * P> acc;
    acc.zero();
    for (int b = 0; b < d->B; ++b) wave::smooth_backward_sequence<R, N, P>(g, b, *s, acc);
    std::vector<R> row(P * P);
    acc.store(row.data());
    put<R>(io->d_LR, row.data(), P * P);
    return 0;
}
template <typename R, int N, int P, int K>
int welbo_fwd(const kvae_dims* d, const kvae_elbo_fwd* io) {
    ElboFwdArgs<R> g = mk<R>(d, io);
    auto s = std::make_unique<wave::ElboScratch<R, N, P>>();
    for (int b = 0; b < d->B; ++b) wave::elbo_forward_sequence<R, N, P>(g, b, *s);
    return 0;
}
template <typename R, int N, int P, int K>
int welbo_bwd(const kvae_dims* d, const kvae_elbo_bwd* io) {
    using PG = ElboParamGrads<N, P>;
    ElboBwdArgs<R> g = mkb<R>(d, io);
    auto s = std::make_unique<wave::ElboBwdScratch<R, N, P>>();
    wave::ElboGradAcc<R, N, P> acc;
    acc.zero();
    for (int b = 0; b < d->B; ++b) wave::elbo_backward_sequence<R, N, P>(g, b, *s, acc);
    std::vector<R> row(PG::LEN);
    acc.store(row.data());
    put<R>(io->d_Q, row.data() + PG::OFF_Q, N * N);
    put<R>(io->d_R, row.data() + PG::OFF_R, P * P);
    put<R>(io->d_m0, row.data() + PG::OFF_M0, N);
    put<R>(io->d_P0, row.data() + PG::OFF_P0, N * N);
    return 0;
}
}  // namespace


// ---- row form (kalman_row.h): the 16 lanes of a sequence as an array ---------------------
namespace {
template <typename R, bool GRAD>
struct HostRowEnv {
    using V = row::Row<R>;
    using SL = row::Slots<GRAD>;
    using Loc = R*;
    int T;
    std::vector<R> lds;
    explicit HostRowEnv(int T_) : T(T_), lds((size_t)T_ * SL::STEP + row::SCRATCH * 16, R(0)) {}
    V scalar(const R* p) const { return V(*p); }
    void fence() const {}
    template <int LEN>
    V vec_c(const R* p) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = (l & 3) < LEN ? p[l & 3] : R(0);
        return x;
    }
    template <int LEN>
    V vec_r(const R* p) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = (l >> 2) < LEN ? p[l >> 2] : R(0);
        return x;
    }
    template <int ROWS, int COLS, int LAY, bool PAD>
    V mat(const R* p) const {
        V x;
        for (int l = 0; l < 16; ++l) {
            const int r = row::lay_r(LAY, l >> 2, l & 3), c = row::lay_c(LAY, l >> 2, l & 3);
            x.v[l] = (r < ROWS && c < COLS) ? p[r * COLS + c] : (PAD && r == c ? R(1) : R(0));
        }
        return x;
    }
    void put_scalar(R* p, const V& x) const { *p = x.v[0]; }
    Loc at_mat(int t, int slot) { return lds.data() + (size_t)t * SL::STEP + slot; }
    Loc at_vec(int t, int slot) { return lds.data() + (size_t)t * SL::STEP + slot; }
    Loc scr(int i) { return lds.data() + (size_t)T * SL::STEP + 16 * i; }
    void st_mat(Loc q, const V& x) const {
        for (int l = 0; l < 16; ++l) q[l] = x.v[l];
    }
    void st_mat_t(Loc q, const V& x) const {
        for (int l = 0; l < 16; ++l) q[4 * (l & 3) + (l >> 2)] = x.v[l];
    }
    V ld_mat(Loc q) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = q[l];
        return x;
    }
    V ld_mat_t(Loc q) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = q[4 * (l & 3) + (l >> 2)];
        return x;
    }
    row::Q4<V> ld_rows4(Loc q) const {
        row::Q4<V> y;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 16; ++l) y.v[k].v[l] = q[4 * (l & 3) + k];
        return y;
    }
    row::Q4<V> ld_cols4(Loc q) const {
        row::Q4<V> y;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 16; ++l) y.v[k].v[l] = q[4 * k + (l & 3)];
        return y;
    }
    template <int KK>
    V ld_colc(Loc q) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = q[4 * (l & 3) + KK];
        return x;
    }
    void st_vec(Loc q, const V& xc) const {
        for (int j = 0; j < 4; ++j) q[j] = xc.v[j];
    }
    V ld_vec_c(Loc q) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = q[l & 3];
        return x;
    }
    V ld_vec_r(Loc q) const {
        V x;
        for (int l = 0; l < 16; ++l) x.v[l] = q[l >> 2];
        return x;
    }
};

template <typename R, int N, int P, int K, bool GRAD, bool SYM>
int row_loss_sym(const kvae_dims* d, const kvae_loss_fwd* fio, const kvae_loss_bwd* bio) {
    using V = row::Row<R>;
    LossBwdArgs<R> g;
    g.f = mkl<R>(d, fio);
    g.burn_in = d->burn_in;
    g.g_scale = bio ? (const R*)bio->g_scale : nullptr;
    g.g_scale_stride = bio ? bio->g_scale_stride : 0;
    g.w_obs = (R)fio->w_obs; g.w_trans = (R)fio->w_trans; g.w_post = (R)fio->w_post;
    g.d_a = bio ? (R*)bio->d_a : nullptr; g.d_alpha = bio ? (R*)bio->d_alpha : nullptr;
    g.partial = nullptr;
    V sums[3] = {V(R(0)), V(R(0)), V(R(0))};
    row::RowAcc<V, P, K> acc;
    row::zero_acc(acc);
    int st = 0;
    for (int b = 0; b < d->B; ++b) {
        HostRowEnv<R, GRAD> env(d->T);
        row::loss_row_sequence<V, R, N, P, K, GRAD, SYM>(env, g, (long)b, sums, acc, st);
    }
    if (st && fio->status) *fio->status |= st;
    if (fio->out) {
        R* out = (R*)fio->out;
        out[0] = (R)fio->w_obs * sums[0].v[0];
        out[1] = (R)fio->w_trans * sums[1].v[0];
        out[2] = (R)fio->w_post * sums[2].v[0];
        out[3] = out[0] + out[1] + out[2];
    }
    if (GRAD) {
        auto mat_out = [&](void* dst, const V& nat, const V* trn, int rows, int cols) {
            if (!dst) return;
            R* o = (R*)dst;
            for (int i = 0; i < rows; ++i)
                for (int j = 0; j < cols; ++j)
                    o[i * cols + j] = nat.v[4 * i + j] + (trn ? trn->v[4 * j + i] : R(0));
        };
        for (int k = 0; k < K; ++k) {
            mat_out(bio->d_A_K ? (R*)bio->d_A_K + k * N * N : nullptr, acc.AK[k], &acc.AKT[k], N, N);
            if (bio->d_C_K)
                for (int r = 0; r < P; ++r)
                    for (int j = 0; j < N; ++j)
                        ((R*)bio->d_C_K)[(k * P + r) * N + j] = acc.CKc[k][r].v[j];
        }
        mat_out(bio->d_Q, acc.Q, nullptr, N, N);
        if (bio->d_R)
            for (int r = 0; r < P; ++r)
                for (int c = 0; c < P; ++c) ((R*)bio->d_R)[r * P + c] = acc.Rs[r][c].v[0];
        mat_out(bio->d_P0, acc.P0, nullptr, N, N);
        if (bio->d_m0)
            for (int j = 0; j < N; ++j) ((R*)bio->d_m0)[j] = acc.m0c.v[j];
    }
    return 0;
}
template <typename R, int N, int P, int K, bool GRAD>
int row_loss(const kvae_dims* d, const kvae_loss_fwd* fio, const kvae_loss_bwd* bio) {
    return d->symmetrize ? row_loss_sym<R, N, P, K, GRAD, true>(d, fio, bio)
                         : row_loss_sym<R, N, P, K, GRAD, false>(d, fio, bio);
}
template <typename R, int N, int P, int K>
int row_loss_fwd(const kvae_dims* d, const kvae_loss_fwd* io) {
    return row_loss<R, N, P, K, false>(d, io, nullptr);
}
template <typename R, int N, int P, int K>
int row_loss_bwd(const kvae_dims* d, const kvae_loss_bwd* io) {
    kvae_loss_fwd f = io->fwd;
    f.out = nullptr;
    f.terms = nullptr;
    f.status = nullptr;
    return row_loss<R, N, P, K, true>(d, &f, io);
}
}  // namespace

#define ROW_DIM_LIST(X) X(4, 2, 3) X(2, 1, 2) X(3, 2, 2) X(4, 2, 8)

#define WAVE_DIM_LIST(X) X(32, 16, 8) X(16, 8, 4) X(6, 3, 2) X(4, 2, 3)

#define DIM_LIST(X) X(4, 2, 3) X(2, 1, 2) X(3, 2, 2) X(8, 4, 4) X(4, 2, 8)
#define CASE(n_, p_, K_)                                          \
    if (d->n == n_ && d->p == p_ && d->K == K_) {                 \
        if (d->dtype == KVAE_F32) return FN<float, n_, p_, K_>ARGS; \
        return FN<double, n_, p_, K_>ARGS;                        \
    }

extern "C" {
int hostsim_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io) {
#define FN filter_fwd
#define ARGS (d, io)
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_filter_forward_masked_mlp(const kvae_dims* d, const kvae_filter_fwd* io,
                                      const kvae_masked_mlp* net) {
#define FN masked_mlp_fwd
#undef ARGS
#define ARGS (d, io, net)
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
#undef ARGS
#define ARGS (d, io)
}
int hostsim_filter_backward_masked_mlp(const kvae_dims* d, const kvae_filter_bwd* io,
                                       const kvae_masked_mlp_bwd* nb) {
#define FN masked_mlp_bwd
#undef ARGS
#define ARGS (d, io, nb)
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
#undef ARGS
#define ARGS (d, io)
}
int hostsim_filter_smooth_fused(const kvae_dims* d, const kvae_filter_fwd* io,
                                const kvae_masked_mlp* net, const kvae_smooth_fwd* s) {
#define FN filter_smooth
#undef ARGS
#define ARGS (d, io, net, s)
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
#undef ARGS
#define ARGS (d, io)
}
int hostsim_predict_future(const kvae_dims* d, const kvae_rollout* io) {
#define FN rollout
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io) {
#define FN filter_bwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io) {
#define FN smooth_fwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io) {
#define FN smooth_bwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io) {
#define FN elbo_fwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io) {
#define FN elbo_bwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_kalman_loss_forward(const kvae_dims* d, const kvae_loss_fwd* io) {
#define FN loss_fwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_kalman_loss_backward(const kvae_dims* d, const kvae_loss_bwd* io) {
#define FN loss_bwd
    DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}

int hostsim_row_kalman_loss_forward(const kvae_dims* d, const kvae_loss_fwd* io) {
#define FN row_loss_fwd
    ROW_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_row_kalman_loss_backward(const kvae_dims* d, const kvae_loss_bwd* io) {
#define FN row_loss_bwd
    ROW_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}

#undef ARGS
#define ARGS (d, io)
int hostsim_wave_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io) {
#define FN wfilter_fwd
    WAVE_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_wave_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io) {
#define FN wfilter_bwd
    WAVE_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_wave_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io) {
#define FN wsmooth_fwd
    WAVE_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_wave_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io) {
#define FN wsmooth_bwd
    WAVE_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_wave_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io) {
#define FN welbo_fwd
    WAVE_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
int hostsim_wave_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io) {
#define FN welbo_bwd
    WAVE_DIM_LIST(CASE)
    return KVAE_E_UNSUPPORTED;
#undef FN
}
}
