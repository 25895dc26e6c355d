// kvae_predict_future: the open-loop rollout of kalman_rollout.h, one lane per sequence.
// With a two-layer perceptron weight network all new steps are ONE launch (the packed
// network is read with scalar loads); with a recurrent network the caller
// runs the network between a one-step EMIT launch and a one-step ADVANCE launch.
#include "launch_common.h"
#include "kalman_rollout.h"

namespace kvae {

constexpr int ROLLOUT_NET_MAX_BYTES = 64 * 1024;
constexpr int SEQ_BLOCK = 64;   // one wavefront per workgroup, one lane per sequence
inline unsigned seq_blocks(long B) { return (unsigned)((B + SEQ_BLOCK - 1) / SEQ_BLOCK); }

template <typename R, int N, int P, int K>
__global__ void __launch_bounds__(SEQ_BLOCK)
rollout_kernel(RolloutArgs<R> g, const R* __restrict__ net) {
    // read-only, non-aliased: the (uniform) loads of the network are scalar loads
    g.net = net;
    const long b = (long)blockIdx.x * SEQ_BLOCK + threadIdx.x;
    if (b < g.B) rollout_sequence<R, N, P, K>(g, (int)b);
}

template <typename R, int N, int P, int K>
int run_rollout(const kvae_dims* d, const kvae_rollout* io, hipStream_t s) {
    if (io->phase == ROLLOUT_ALL &&
        (size_t)(io->hidden * (P + 1 + K) + K) * sizeof(R) > (size_t)ROLLOUT_NET_MAX_BYTES)
        return KVAE_E_UNSUPPORTED;
    RolloutArgs<R> g;
    g.S = d->T; g.B = d->B; g.hidden = io->hidden; g.phase = io->phase; g.sample = io->sample;
    g.A_K = (const R*)io->A_K; g.C_K = (const R*)io->C_K; g.Q = (const R*)io->Q;
    g.mn_in = (const R*)io->mean_pred; g.Pn_in = (const R*)io->cov_pred;
    g.C_in = (const R*)io->mat_C; g.eps = (const R*)io->eps; g.net = nullptr;
    g.alpha = (const R*)io->alpha;
    g.as_new = (R*)io->as_new; g.means = (R*)io->means; g.covs = (R*)io->covs;
    g.mn_new = (R*)io->mean_pred_new; g.Pn_new = (R*)io->cov_pred_new;
    g.As_new = (R*)io->mat_As_new; g.Cs_new = (R*)io->mat_Cs_new; g.status = io->status;
    hipLaunchKernelGGL((rollout_kernel<R, N, P, K>), dim3((unsigned)seq_blocks(d->B)),
                       dim3(SEQ_BLOCK), 0, s, g, (const R*)io->packed_mlp);
    return launch_status();
}

}  // namespace kvae

using namespace kvae;

#define KVAE_DISPATCH_CASE(n_, p_, K_)                                  \
    if (d->n == n_ && d->p == p_ && d->K == K_) {                       \
        if (d->dtype == KVAE_F32) return KVAE_CALL(float, n_, p_, K_);  \
        return KVAE_CALL(double, n_, p_, K_);                           \
    }

extern "C" int kvae_predict_future(const kvae_dims* d, const kvae_rollout* io, void* stream) {
    if (!dims_ok(d) || !io || !io->A_K || !io->C_K || !io->Q || !io->mean_pred || !io->cov_pred)
        return KVAE_E_BADARG;
    const bool emits = io->phase != KVAE_ROLLOUT_ADVANCE, advances = io->phase != KVAE_ROLLOUT_EMIT;
    if (io->phase < KVAE_ROLLOUT_ALL || io->phase > KVAE_ROLLOUT_ADVANCE) return KVAE_E_BADARG;
    if (io->phase != KVAE_ROLLOUT_ALL && d->T != 1) return KVAE_E_BADARG;
    if (emits && (!io->mat_C || !io->as_new || !io->means || !io->covs)) return KVAE_E_BADARG;
    if (emits && io->sample && !io->eps) return KVAE_E_BADARG;
    if (advances && (!io->mat_As_new || !io->mat_Cs_new || !io->mean_pred_new ||
                     !io->cov_pred_new))
        return KVAE_E_BADARG;
    if (io->phase == KVAE_ROLLOUT_ALL && (!io->packed_mlp || io->hidden <= 0)) return KVAE_E_BADARG;
    if (io->phase == KVAE_ROLLOUT_ADVANCE && !io->alpha) return KVAE_E_BADARG;
#define KVAE_CALL(R, n_, p_, K_) run_rollout<R, n_, p_, K_>(d, io, (hipStream_t)stream)
    KVAE_DIM_LIST(KVAE_DISPATCH_CASE)
#undef KVAE_CALL
    return KVAE_E_UNSUPPORTED;
}
