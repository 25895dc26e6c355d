// Filter and RTS smoother of one sequence in ONE pass over the data, for consumers that
// need only some of the ten tensors the two separate calls produce (imputation reads the
// imputed and the re-sampled observations: kvae/kalman_vae.py:141-160,
// kvae/evaluation.py:100-140).  One lane per sequence: the forward sweep is the filter
// loop of kalman_core.h (optionally with the perceptron weight network in the loop), the
// reverse sweep is the statements of smooth_forward_sequence with the per-step matrices
// re-mixed from the weights and the one-step predictions recomputed from the filtered
// moments (m' = A m, P' = sym(A P A^T + Q), the filter's own time update) instead of read
// back, and every output pointer may be null.  The filtered moments the reverse sweep needs
// go through buffers that are the caller's outputs where requested and a workspace
// otherwise; a lane reads back only what it wrote itself, so the two sweeps need no
// synchronisation.
#pragma once

#include "kalman_core.h"

namespace kvae {

template <typename R>
struct FilterSmoothArgs {
    FilterFwdArgs<R> f;     // m, Pf: never null (outputs or workspace); mn, Pn, As, Cs may be
    const R *LR, *eps_s, *eps_o;
    R *ms, *Ps, *zs, *ares;   // each may be null
};

template <typename R, int N, int P>
KV_HD bool filter_smooth_emit(const FilterSmoothArgs<R>& g, long tb, const R* mean, const R* cov,
                              const R* C, const R* LR) {
    if (!g.zs && !g.ares) return true;
    bool ok = true;
    R z[N], ar[P];
    vcopy<R, N>(mean, z);
    if (g.eps_s) {
        R L[N * N], e[N];
        ok = cholesky<R, N>(cov, L);
        gload<R, N>(g.eps_s + tb * N, e);
        gemv<R, N, N, false, 1>(L, e, z);
    }
    gemv<R, P, N, false, 0>(C, z, ar);
    if (g.eps_o) {
        R e[P];
        gload<R, P>(g.eps_o + tb * P, e);
        gemv<R, P, P, false, 1>(LR, e, ar);
    }
    if (g.zs) gstore<R, N>(g.zs + tb * N, z);
    if (g.ares) gstore<R, P>(g.ares + tb * P, ar);
    return ok;
}

// q: the perceptron network evaluated in the filter loop, or null (weights read from f.alpha)
template <typename R, int N, int P, int K>
KV_HD void filter_smooth_sequence(const FilterSmoothArgs<R>& g, const MaskedNetArgs<R>* q, int b) {
    const FilterFwdArgs<R>& f = g.f;
    if (q) {
        filter_forward_masked_mlp_sequence<R, N, P, K>(f, *q, b);
    } else {
        filter_forward_sequence<R, N, P, K>(f, b);
    }

    const long B = f.B;
    const int T = f.T;
    R LR[P * P];
    if (g.eps_o) {
        gload<R, P * P>(g.LR, LR);
    } else {
        vzero<R, P * P>(LR);
    }
    R Qm[N * N];
    gload<R, N * N>(f.Q, Qm);
    int st = 0;
    R ms[N], Ps[N * N], C[P * N], w[K], wn[K];
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(f.m + tb * N, ms);
        gload<R, N * N>(f.Pf + tb * N * N, Ps);
        gload<R, K>(f.alpha + (tb + B) * K, wn);
        mix<R, K, P * N>(wn, f.C_K, C);   // C_T, as the smoother's last step uses it
        if (g.ms) gstore<R, N>(g.ms + tb * N, ms);
        if (g.Ps) gstore<R, N * N>(g.Ps + tb * N * N, Ps);
        if (!filter_smooth_emit<R, N, P>(g, tb, ms, Ps, C, LR)) st |= ST_NONPD_SMOOTHED;
    }
    for (int t = T - 2; t >= 0; --t) {
        const long tb = (long)t * B + b;
        R m[N], Pf[N * N], mn[N], Pn[N * N], An[N * N];
        gload<R, N>(f.m + tb * N, m);
        gload<R, N * N>(f.Pf + tb * N * N, Pf);
        gload<R, K>(f.alpha + (tb + B) * K, wn);
        gload<R, K>(f.alpha + tb * K, w);
        mix<R, K, N * N>(wn, f.A_K, An);
        mix<R, K, P * N>(w, f.C_K, C);
        {
            // the time update of filter_forward_loop, again
            R T0[N * N];
            gemv<R, N, N, false, 0>(An, m, mn);
            gemm<R, N, N, N, false, false, 0>(An, Pf, T0);
            vcopy<R, N * N>(Qm, Pn);
            gemm<R, N, N, N, false, true, 1>(T0, An, Pn);
            if (f.symmetrize) symmetrize_inplace<R, N>(Pn);
        }

        R Pi[N * N], W[N * N], J[N * N];
        if (!spd_inverse<R, N>(Pn, Pi)) st |= ST_NONPD_PREDICTED;
        gemm<R, N, N, N, true, false, 0>(An, Pi, W);   // An^T Pi
        gemm<R, N, N, N, false, false, 0>(Pf, W, J);   // J = Pf An^T Pi
        R d[N], D[N * N], T1[N * N];
        KV_UNROLL
        for (int i = 0; i < N; ++i) d[i] = ms[i] - mn[i];
        KV_UNROLL
        for (int i = 0; i < N * N; ++i) D[i] = Ps[i] - Pn[i];
        vcopy<R, N>(m, ms);
        gemv<R, N, N, false, 1>(J, d, ms);
        gemm<R, N, N, N, false, false, 0>(J, D, T1);
        vcopy<R, N * N>(Pf, Ps);
        gemm<R, N, N, N, false, true, 1>(T1, J, Ps);
        if (f.symmetrize) symmetrize_inplace<R, N>(Ps);
        if (g.ms) gstore<R, N>(g.ms + tb * N, ms);
        if (g.Ps) gstore<R, N * N>(g.Ps + tb * N * N, Ps);
        if (!filter_smooth_emit<R, N, P>(g, tb, ms, Ps, C, LR)) st |= ST_NONPD_SMOOTHED;
    }
    if (st && f.status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(f.status, st);
#else
        *f.status |= st;
#endif
    }
}

}  // namespace kvae
