// Open-loop rollout past the end of a filtered sequence, one lane per sequence
// (kvae/state_space_model.py:486-575).  Per new step, from the last one-step prediction
// (mn, Pn) and the last mixed observation matrix C:
//   emit:     means <- mn, covs <- Pn;  z = mn (+ chol(Pn) eps: "sample");  a = C z
//   weights:  w = net(a)   (two-layer perceptron in the loop, or supplied by the caller
//             who steps a recurrent network between `emit` and `advance`)
//   advance:  A = sum_k w_k A_k, C = sum_k w_k C_k;  mn <- A mn;
//             Pn <- sym(Q) ("sample", :550-551) or sym(A Pn A^T + Q) ("mean")
// Host-compilable (KV_HD) so that CPU tests can run the same statements.
#pragma once

#include "kalman_core.h"

namespace kvae {

enum RolloutPhase { ROLLOUT_ALL = 0, ROLLOUT_EMIT = 1, ROLLOUT_ADVANCE = 2 };

template <typename R>
struct RolloutArgs {
    int S;            // new steps written by this call (1 for EMIT / ADVANCE)
    long B;
    int hidden, phase, sample;
    const R *A_K, *C_K, *Q;
    const R *mn_in, *Pn_in, *C_in;   // (B,N), (B,N,N), (B,P,N)
    const R* eps;                    // (S,B,N) or null
    const R* net;                    // packed perceptron (ALL) or null
    const R* alpha;                  // (B,K) weights of the new step (ADVANCE)
    R *as_new, *means, *covs, *mn_new, *Pn_new, *As_new, *Cs_new;
    int* status;
};

template <typename R, int N, int P, int K>
KV_HD void rollout_sequence(const RolloutArgs<R>& g, int b) {
    const long B = g.B;
    R mn[N], Pn[N * N], C[P * N];
    gload<R, N>(g.mn_in + (long)b * N, mn);
    gload<R, N * N>(g.Pn_in + (long)b * N * N, Pn);
    if (g.phase != ROLLOUT_ADVANCE) gload<R, P * N>(g.C_in + (long)b * P * N, C);
    R Q[N * N];
    gload<R, N * N>(g.Q, Q);
    int st = 0;
    for (int s = 0; s < g.S; ++s) {
        const long sb = (long)s * B + b;
        R w[K];
        if (g.phase != ROLLOUT_ADVANCE) {
            gstore<R, N>(g.means + sb * N, mn);
            gstore<R, N * N>(g.covs + sb * N * N, Pn);
            R z[N], a[P];
            vcopy<R, N>(mn, z);
            if (g.eps) {
                R L[N * N], e[N];
                if (!cholesky<R, N>(Pn, L)) st |= ST_NONPD_PREDICTED;
                gload<R, N>(g.eps + sb * N, e);
                gemv<R, N, N, false, 1>(L, e, z);
            }
            gemv<R, P, N, false, 0>(C, z, a);
            gstore<R, P>(g.as_new + sb * P, a);
            if (g.phase == ROLLOUT_EMIT) break;
            mlp2_softmax<R, P, K>(g.net, g.hidden, a, w);
        } else {
            gload<R, K>(g.alpha + (long)b * K, w);
        }
        R A[N * N], T1[N * N], m2[N];
        mix<R, K, N * N>(w, g.A_K, A);
        mix<R, K, P * N>(w, g.C_K, C);
        gstore<R, N * N>(g.As_new + sb * N * N, A);
        gstore<R, P * N>(g.Cs_new + sb * P * N, C);
        gemv<R, N, N, false, 0>(A, mn, m2);
        vcopy<R, N>(m2, mn);
        if (g.sample) {
            vcopy<R, N * N>(Q, Pn);
        } else {
            gemm<R, N, N, N, false, false, 0>(A, Pn, T1);
            vcopy<R, N * N>(Q, Pn);
            gemm<R, N, N, N, false, true, 1>(T1, A, Pn);
        }
        symmetrize_inplace<R, N>(Pn);
        gstore<R, N>(g.mn_new + sb * N, mn);
        gstore<R, N * N>(g.Pn_new + sb * N * N, Pn);
    }
    if (st && g.status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(g.status, st);
#else
        *g.status |= st;
#endif
    }
}

}  // namespace kvae
