// The Kalman part of the training objective as ONE pass per direction over a
// sequence: filter -> RTS smoother -> sample z ~ N(smoothed) -> the three
// log-density terms, and the adjoint of all of it.  One lane owns one sequence
// (execution model and helpers: kalman_core.h).
//
// What the training call site consumes (reference: kvae/kalman_vae.py:141-260) is
// three scalars, aggregate(obs), aggregate(trans), aggregate(post); the twelve
// filter / smoother tensors and the per-step mixed matrices are only intermediate
// values there.  The forward pass below therefore writes, per sequence and step,
// only the state the backward pass reads again (filtered and predicted moments,
// smoothed moments: `saved`), and sums the terms as it goes; the backward pass
// writes d_a and d_alpha and per-wavefront partial sums of the parameter
// gradients.  mat_As / mat_Cs never exist in memory: A_t, C_t are re-mixed from
// alpha where they are used, and their adjoints are contracted with the bases
// on the spot.
//
// Forward, per sequence:
//   phase F (t = 0..T-1):  filter_forward_sequence          state_space_model.py:284-352
//   phase S (t = T-1..0):  RTS step, z_t = ms_t + chol(Ps_t) eps_t, obs_t, post_t and
//                          trans_{t+1} (needs z_t and z_{t+1})   state_space_model.py:424-446,
//                                                                 kalman_vae.py:191-242
// Backward, per sequence:
//   phase SA (t = 0..T-1): adjoint of the terms of step t and of the RTS step t
//                          (carry: gradient w.r.t. the smoothed moments of step t+1);
//                          leaves the gradients w.r.t. the filter's four outputs in
//                          the workspace and the first part of d_a, d_alpha
//   phase FA (t = T-1..0): filter_backward_core, adding to d_a, d_alpha
#pragma once

#include "kalman_core.h"

namespace kvae {

template <typename R>
struct LossFwdArgs {
    int T, B, symmetrize;
    const R *a, *alpha, *mask, *A_K, *C_K, *Q, *Rm, *m0, *P0, *eps;
    // saved for the backward pass, written here: (T,B,n), (T,B,n,n) each
    R *m, *Pf, *mn, *Pn, *ms, *Ps;
    R* terms;    // (3,T,B) obs, trans, post per pair, or null
    R* partial;  // (rows, 3) per-wavefront sums of obs, trans, post
    int* status;
};

template <typename R>
struct LossBwdArgs {
    LossFwdArgs<R> f;   // same pointers as the forward call (saved tensors are read)
    int burn_in;
    // cotangent of every (t,b) entry of a term: weight * g_scale[k * stride]
    const R* g_scale;
    int g_scale_stride;
    R w_obs, w_trans, w_post;
    R *d_a, *d_alpha;                      // (T,B,p), (T+1,B,K)
    R *bar_m, *bar_P, *bar_mn, *bar_Pn;    // workspace, shapes of the filter outputs
    R* partial;                            // (rows, LossParamGrads::LEN)
};

// Parameter-gradient sums of the fused backward; the first four blocks are laid out
// like FilterParamGrads so that filter_backward_core adds to the same array.
template <int N, int P, int K>
struct LossParamGrads {
    using FG = FilterParamGrads<N, P, K>;
    static constexpr int OFF_A = FG::OFF_A;
    static constexpr int OFF_C = FG::OFF_C;
    static constexpr int OFF_Q = FG::OFF_Q;
    static constexpr int OFF_R = FG::OFF_R;
    static constexpr int OFF_M0 = FG::LEN;
    static constexpr int OFF_P0 = OFF_M0 + N;
    static constexpr int LEN = OFF_P0 + N * N;
};

// Layout of the opaque `saved` buffer of the C ABI: [m | Pf | mn | Pn | ms | Ps], and of
// the backward's workspace head: [bar_m | bar_P | bar_mn | bar_Pn].
inline long loss_saved_elems(long T, long B, long n) { return T * B * (3 * n + 3 * n * n); }
inline long loss_bars_elems(long T, long B, long n) { return T * B * (2 * n + 2 * n * n); }

template <typename R>
inline void loss_saved_pointers(void* saved, long T, long B, long n, LossFwdArgs<R>& g) {
    R* p = static_cast<R*>(saved);
    const long v = T * B * n, c = T * B * n * n;
    g.m = p;
    g.Pf = g.m + v;
    g.mn = g.Pf + c;
    g.Pn = g.mn + v;
    g.ms = g.Pn + c;
    g.Ps = g.ms + v;
}
template <typename R>
inline void loss_bars_pointers(void* ws, long T, long B, long n, LossBwdArgs<R>& g) {
    R* p = static_cast<R*>(ws);
    const long v = T * B * n, c = T * B * n * n;
    g.bar_m = p;
    g.bar_P = g.bar_m + v;
    g.bar_mn = g.bar_P + c;
    g.bar_Pn = g.bar_mn + v;
}

template <typename R>
KV_HD void raise_status(int* status, int st) {
    if (st && status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(status, st);
#else
        *status |= st;
#endif
    }
}

// sum of log of the diagonal of a lower factor
template <typename R, int D>
KV_HD R half_logdet(const R* L) {
    R s = R(0);
    KV_UNROLL
    for (int i = 0; i < D; ++i) s += log_pos(L[i * D + i]);
    return s;
}

// -1/2 (D log 2pi + |L^-1 r|^2) - hl ; r is overwritten by L^-1 r
template <typename R, int D>
KV_HD R gauss_logpdf_hl(const R* L, R hl, R* r) {
    solve_lower<R, D>(L, r);
    return R(-0.5) * (R(D) * log_2pi<R>() + vdot<R, D>(r, r)) - hl;
}

// S^-1 from the lower factor
template <typename R, int D>
KV_HD void inverse_from_factor(const R* L, R* Si) {
    R Li[D * D];
    tri_lower_inverse<R, D>(L, Li);
    gemm<R, D, D, D, true, false, 0>(Li, Li, Si);
}

// ---------------------------------------------------------------------------
// Forward.  sums[0..2] += obs, trans, post of this sequence summed over t.
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
KV_HD void loss_forward_sequence(const LossFwdArgs<R>& g, int b, R* sums) {
    const long B = g.B;
    const int T = g.T;
    {
        FilterFwdArgs<R> f;
        f.T = T; f.B = g.B; f.symmetrize = g.symmetrize; f.carry_per_sequence = 0;
        f.a = g.a; f.alpha = g.alpha; f.mask = g.mask; f.A_K = g.A_K; f.C_K = g.C_K;
        f.Q = g.Q; f.Rm = g.Rm; f.m_in = g.m0; f.P_in = g.P0;
        f.m = g.m; f.Pf = g.Pf; f.mn = g.mn; f.Pn = g.Pn; f.As = nullptr; f.Cs = nullptr;
        f.status = g.status;
        filter_forward_sequence<R, N, P, K>(f, b);
    }
    int st = 0;
    R LR[P * P], LQ[N * N];
    {
        R Rm[P * P], Q[N * N];
        gload<R, P * P>(g.Rm, Rm);
        gload<R, N * N>(g.Q, Q);
        if (!cholesky<R, P>(Rm, LR)) st |= ST_NONPD_NOISE;
        if (!cholesky<R, N>(Q, LQ)) st |= ST_NONPD_NOISE;
    }
    const R hlR = half_logdet<R, P>(LR), hlQ = half_logdet<R, N>(LQ);
    R s_obs = R(0), s_trans = R(0), s_post = R(0);
    R* t_obs = g.terms;
    R* t_trans = g.terms ? g.terms + (long)T * B : nullptr;
    R* t_post = g.terms ? g.terms + 2 * (long)T * B : nullptr;

    // z_t, obs_t and post_t from the smoothed moments of step t
    auto emit = [&](long tb, const R* ms, const R* Ps, R* z) {
        R L[N * N], e[N], w[K], C[P * N], r[P];
        if (!cholesky<R, N>(Ps, L)) st |= ST_NONPD_SMOOTHED;
        gload<R, N>(g.eps + tb * N, e);
        vcopy<R, N>(ms, z);
        gemv<R, N, N, false, 1>(L, e, z);
        const R post = R(-0.5) * (R(N) * log_2pi<R>() + vdot<R, N>(e, e)) - half_logdet<R, N>(L);
        gload<R, K>(g.alpha + tb * K, w);
        mix<R, K, P * N>(w, g.C_K, C);
        gload<R, P>(g.a + tb * P, r);
        gemv<R, P, N, false, 2>(C, z, r);
        const R obs = gauss_logpdf_hl<R, P>(LR, hlR, r);
        s_obs += obs;
        s_post += post;
        if (t_obs) {
            t_obs[tb] = obs;
            t_post[tb] = post;
        }
    };

    R ms[N], Ps[N * N], zn[N];
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(g.m + tb * N, ms);
        gload<R, N * N>(g.Pf + tb * N * N, Ps);
        gstore<R, N>(g.ms + tb * N, ms);
        gstore<R, N * N>(g.Ps + tb * N * N, Ps);
        emit(tb, ms, Ps, zn);
    }
    for (int t = T - 2; t >= 0; --t) {
        const long tb = (long)t * B + b;
        R m[N], Pf[N * N], mn[N], Pn[N * N], An[N * N], wn[K];
        gload<R, N>(g.m + tb * N, m);
        gload<R, N * N>(g.Pf + tb * N * N, Pf);
        gload<R, N>(g.mn + tb * N, mn);
        gload<R, N * N>(g.Pn + tb * N * N, Pn);
        gload<R, K>(g.alpha + (tb + B) * K, wn);
        mix<R, K, N * N>(wn, g.A_K, An);

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
        if (g.symmetrize) symmetrize_inplace<R, N>(Ps);
        gstore<R, N>(g.ms + tb * N, ms);
        gstore<R, N * N>(g.Ps + tb * N * N, Ps);

        R z[N];
        emit(tb, ms, Ps, z);
        // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q)
        R r[N];
        vcopy<R, N>(zn, r);
        gemv<R, N, N, false, 2>(An, z, r);
        const R tr = gauss_logpdf_hl<R, N>(LQ, hlQ, r);
        s_trans += tr;
        if (t_trans) t_trans[tb + B] = tr;
        vcopy<R, N>(z, zn);
    }
    {
        // trans_0 = log N(z_0; m0, P0)
        R P0[N * N], L0[N * N], r[N], m0[N];
        gload<R, N * N>(g.P0, P0);
        gload<R, N>(g.m0, m0);
        if (!cholesky<R, N>(P0, L0)) st |= ST_NONPD_NOISE;
        KV_UNROLL
        for (int i = 0; i < N; ++i) r[i] = zn[i] - m0[i];
        const R tr = gauss_logpdf_hl<R, N>(L0, half_logdet<R, N>(L0), r);
        s_trans += tr;
        if (t_trans) t_trans[b] = tr;
    }
    sums[0] += s_obs;
    sums[1] += s_trans;
    sums[2] += s_post;
    raise_status<R>(g.status, st);
}

// ---------------------------------------------------------------------------
// Backward.  acc: LossParamGrads<N,P,K>::LEN running sums of this lane.
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
KV_HD void loss_backward_sequence(const LossBwdArgs<R>& g, int b, R* acc) {
    using PG = LossParamGrads<N, P, K>;
    const LossFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const R wo = g.w_obs * g.g_scale[0];
    const R wt = g.w_trans * g.g_scale[g.g_scale_stride];
    const R wp = g.w_post * g.g_scale[2 * g.g_scale_stride];

    R LR[P * P], LQ[N * N];
    {
        R Rm[P * P], Q[N * N];
        gload<R, P * P>(f.Rm, Rm);
        gload<R, N * N>(f.Q, Q);
        cholesky<R, P>(Rm, LR);
        cholesky<R, N>(Q, LQ);
    }

    // ---- phase SA: terms + smoother adjoint, forward in time -------------------
    // sample of a step: z = ms + L eps with L = chol(Ps)
    auto sample = [&](long tb, R* z, R* L, R* e) {
        R ms[N], Ps[N * N];
        gload<R, N>(f.ms + tb * N, ms);
        gload<R, N * N>(f.Ps + tb * N * N, Ps);
        gload<R, N>(f.eps + tb * N, e);
        cholesky<R, N>(Ps, L);
        vcopy<R, N>(ms, z);
        gemv<R, N, N, false, 1>(L, e, z);
    };

    R z[N], L[N * N], e[N];   // step t
    sample((long)b, z, L, e);
    R ut[N];                  // Sigma^-1 (z_t - mean) of the transition term whose x is z_t
    R suu_Q[N * N];           // sum of u u^T over the Q-transition terms (t >= 1)
    R suu_R[P * P];           // sum of u u^T over the observation terms
    vzero<R, N * N>(suu_Q);
    vzero<R, P * P>(suu_R);
    {
        // trans_0: x = z_0, mean m0, covariance P0
        R P0[N * N], L0[N * N], m0[N], P0i[N * N];
        gload<R, N * N>(f.P0, P0);
        gload<R, N>(f.m0, m0);
        cholesky<R, N>(P0, L0);
        KV_UNROLL
        for (int i = 0; i < N; ++i) ut[i] = z[i] - m0[i];
        solve_lower<R, N>(L0, ut);
        solve_lower_t<R, N>(L0, ut);
        inverse_from_factor<R, N>(L0, P0i);
        KV_UNROLL
        for (int i = 0; i < N; ++i) acc[PG::OFF_M0 + i] += wt * ut[i];
        KV_UNROLL
        for (int i = 0; i < N; ++i) {
            KV_UNROLL
            for (int j = 0; j < N; ++j)
                acc[PG::OFF_P0 + i * N + j] += wt * R(0.5) * (ut[i] * ut[j] - P0i[i * N + j]);
        }
    }

    R msb[N], Psb[N * N];   // carry: gradient w.r.t. the smoothed moments of step t
    vzero<R, N>(msb);
    vzero<R, N * N>(Psb);
    R dA_row[K];            // A-part of d_alpha[t], produced by step t-1
    vzero<R, K>(dA_row);

    for (int t = 0; t < T; ++t) {
        const long tb = (long)t * B + b;
        const bool last = t == T - 1;
        R w[K], wn[K], C[P * N], An[N * N], a[P];
        gload<R, K>(f.alpha + tb * K, w);
        gload<R, K>(f.alpha + (tb + B) * K, wn);
        gload<R, P>(f.a + tb * P, a);
        mix<R, K, P * N>(w, f.C_K, C);
        mix<R, K, N * N>(wn, f.A_K, An);
        R zn[N], Ln[N * N], en[N];
        if (!last) sample(tb + B, zn, Ln, en);

        // -- adjoint of the terms that read z_t --
        R zb[N];
        R Cb[P * N];
        {
            // obs_t = log N(a_t; C_t z_t, R)
            R u[P];
            vcopy<R, P>(a, u);
            gemv<R, P, N, false, 2>(C, z, u);
            solve_lower<R, P>(LR, u);
            solve_lower_t<R, P>(LR, u);   // R^-1 (a - C z)
            R da[P];
            KV_UNROLL
            for (int i = 0; i < P; ++i) da[i] = -wo * u[i];
            gstore<R, P>(g.d_a + tb * P, da);
            vzero<R, P * N>(Cb);
            outer_add<R, P, N>(wo, u, z, Cb);
            gemv<R, N, P, true, 0>(C, u, zb);   // C^T u
            KV_UNROLL
            for (int i = 0; i < N; ++i) zb[i] = wo * zb[i] - wt * ut[i];
            outer_add<R, P, P>(R(1), u, u, suu_R);
        }
        R Ab[N * N];
        vzero<R, N * N>(Ab);
        R un[N];
        if (!last) {
            // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q)
            vcopy<R, N>(zn, un);
            gemv<R, N, N, false, 2>(An, z, un);
            solve_lower<R, N>(LQ, un);
            solve_lower_t<R, N>(LQ, un);
            R s[N];
            gemv<R, N, N, true, 0>(An, un, s);   // A^T u
            KV_UNROLL
            for (int i = 0; i < N; ++i) zb[i] += wt * s[i];
            outer_add<R, N, N>(wt, un, z, Ab);
            outer_add<R, N, N>(R(1), un, un, suu_Q);
        }
        // z_t = ms_t + L_t eps_t ; post_t = const - sum log diag L_t
        vadd<R, N>(zb, msb);
        {
            R Lb[N * N];
            vzero<R, N * N>(Lb);
            outer_add<R, N, N>(R(1), zb, e, Lb);
            KV_UNROLL
            for (int i = 0; i < N; ++i) Lb[i * N + i] -= wp * recip(L[i * N + i]);
            cholesky_adjoint_add<R, N>(L, Lb, Psb);
        }
        // C-part of d_alpha[t]; basis gradient of C
        {
            R row[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) {
                R sC = R(0);
                KV_UNROLL
                for (int q = 0; q < P * N; ++q) {
                    sC += Cb[q] * f.C_K[k * P * N + q];
                    acc[PG::OFF_C + k * P * N + q] += w[k] * Cb[q];
                }
                row[k] = dA_row[k] + sC;
            }
            gstore<R, K>(g.d_alpha + tb * K, row);
        }

        if (last) {
            // smoothed == filtered at the last step (no burn-in gate there)
            R z0[N * N];
            vzero<R, N * N>(z0);
            gstore<R, N>(g.bar_m + tb * N, msb);
            gstore<R, N * N>(g.bar_P + tb * N * N, Psb);
            gstore<R, N>(g.bar_mn + tb * N, z0);
            gstore<R, N * N>(g.bar_Pn + tb * N * N, z0);
            break;
        }
        if (t < g.burn_in) {
            vzero<R, N>(msb);
            vzero<R, N * N>(Psb);
        }

        // -- adjoint of the RTS step t (statements of smooth_backward_sequence) --
        R m[N], Pf[N * N], mn[N], Pn[N * N], ms1[N], Ps1[N * N];
        gload<R, N>(f.m + tb * N, m);
        gload<R, N * N>(f.Pf + tb * N * N, Pf);
        gload<R, N>(f.mn + tb * N, mn);
        gload<R, N * N>(f.Pn + tb * N * N, Pn);
        gload<R, N>(f.ms + (tb + B) * N, ms1);
        gload<R, N * N>(f.Ps + (tb + B) * N * N, Ps1);
        R Pi[N * N], W[N * N], J[N * N], d[N], D[N * N];
        spd_inverse<R, N>(Pn, Pi);
        gemm<R, N, N, N, true, false, 0>(An, Pi, W);
        gemm<R, N, N, N, false, false, 0>(Pf, W, J);
        KV_UNROLL
        for (int i = 0; i < N; ++i) d[i] = ms1[i] - mn[i];
        KV_UNROLL
        for (int i = 0; i < N * N; ++i) D[i] = Ps1[i] - Pn[i];
        if (f.symmetrize) symmetrize_inplace<R, N>(Psb);  // H
        gstore<R, N>(g.bar_m + tb * N, msb);
        R Pfb[N * N];
        vcopy<R, N * N>(Psb, Pfb);
        R Jb[N * N], T1[N * N];
        vzero<R, N * N>(Jb);
        outer_add<R, N, N>(R(1), msb, d, Jb);
        gemm<R, N, N, N, false, true, 0>(J, D, T1);        // J D^T
        gemm<R, N, N, N, false, false, 1>(Psb, T1, Jb);
        gemm<R, N, N, N, false, false, 0>(J, D, T1);       // J D
        gemm<R, N, N, N, true, false, 1>(Psb, T1, Jb);
        R db[N], Db[N * N];
        gemv<R, N, N, true, 0>(J, msb, db);
        gemm<R, N, N, N, true, false, 0>(J, Psb, T1);
        gemm<R, N, N, N, false, false, 0>(T1, J, Db);
        gemm<R, N, N, N, false, true, 1>(Jb, W, Pfb);      // Pfb += Jb W^T
        R Wb[N * N], Pib[N * N], Pnb[N * N];
        gemm<R, N, N, N, true, false, 0>(Pf, Jb, Wb);      // Pf^T Jb
        gemm<R, N, N, N, false, true, 1>(Pi, Wb, Ab);      // Ab += Pi Wb^T
        gemm<R, N, N, N, false, false, 0>(An, Wb, Pib);    // Pib = An Wb
        gemm<R, N, N, N, true, false, 0>(Pi, Pib, T1);
        gemm<R, N, N, N, false, true, 0>(T1, Pi, Pnb);
        KV_UNROLL
        for (int i = 0; i < N * N; ++i) Pnb[i] = -Pnb[i] - Db[i];
        R mnb[N];
        KV_UNROLL
        for (int i = 0; i < N; ++i) mnb[i] = -db[i];
        gstore<R, N * N>(g.bar_P + tb * N * N, Pfb);
        gstore<R, N>(g.bar_mn + tb * N, mnb);
        gstore<R, N * N>(g.bar_Pn + tb * N * N, Pnb);
        // A-part of d_alpha[t+1]; basis gradient of A
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            R sA = R(0);
            KV_UNROLL
            for (int q = 0; q < N * N; ++q) {
                sA += Ab[q] * f.A_K[k * N * N + q];
                acc[PG::OFF_A + k * N * N + q] += wn[k] * Ab[q];
            }
            dA_row[k] = sA;
        }
        // carry to step t+1; the next step's sample becomes the current one
        vcopy<R, N>(db, msb);
        vcopy<R, N * N>(Db, Psb);
        vcopy<R, N>(zn, z);
        vcopy<R, N * N>(Ln, L);
        vcopy<R, N>(en, e);
        vcopy<R, N>(un, ut);
    }
    // covariance gradients of the Gaussian terms: w/2 (sum u u^T - count * Sigma^-1)
    {
        R Ri[P * P], Qi[N * N];
        inverse_from_factor<R, P>(LR, Ri);
        inverse_from_factor<R, N>(LQ, Qi);
        KV_UNROLL
        for (int i = 0; i < P * P; ++i)
            acc[PG::OFF_R + i] += wo * R(0.5) * (suu_R[i] - R(T) * Ri[i]);
        KV_UNROLL
        for (int i = 0; i < N * N; ++i)
            acc[PG::OFF_Q + i] += wt * R(0.5) * (suu_Q[i] - R(T - 1) * Qi[i]);
    }

    // ---- phase FA: filter adjoint, reverse in time -------------------------------
    FilterBwdArgs<R> fb;
    fb.f.T = T; fb.f.B = f.B; fb.f.symmetrize = f.symmetrize; fb.f.carry_per_sequence = 0;
    fb.f.a = f.a; fb.f.alpha = f.alpha; fb.f.mask = f.mask; fb.f.A_K = f.A_K; fb.f.C_K = f.C_K;
    fb.f.Q = f.Q; fb.f.Rm = f.Rm; fb.f.m_in = f.m0; fb.f.P_in = f.P0;
    fb.f.m = f.m; fb.f.Pf = f.Pf; fb.f.mn = f.mn; fb.f.Pn = f.Pn;
    fb.f.As = nullptr; fb.f.Cs = nullptr; fb.f.status = nullptr;
    fb.burn_in = g.burn_in;
    fb.g_m = g.bar_m; fb.g_P = g.bar_P; fb.g_mn = g.bar_mn; fb.g_Pn = g.bar_Pn;
    fb.g_As = nullptr; fb.g_Cs = nullptr;
    fb.d_a = g.d_a; fb.d_alpha = g.d_alpha; fb.d_m_in = nullptr; fb.d_P_in = nullptr;
    fb.partial = nullptr; fb.bar_As = nullptr; fb.bar_Cs = nullptr;
    R m0b[N], P0b[N * N];
    filter_backward_core<R, N, P, K, true>(fb, b, acc, m0b, P0b);
    vadd<R, N>(m0b, acc + PG::OFF_M0);
    vadd<R, N * N>(P0b, acc + PG::OFF_P0);
}

}  // namespace kvae
