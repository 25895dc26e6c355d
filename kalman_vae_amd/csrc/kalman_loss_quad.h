// Four lanes per sequence ("quad") form of the fused Kalman objective of
// kalman_loss.h: same phases, same statements, on the distributed layout of
// kalman_quad.h (lane q of a quad holds column q and row q of every matrix, vectors
// are replicated in the four lanes, 4x4 products run on v_mfma_f32_4x4x1).  Used while
// one lane per sequence would leave most SIMDs of the chip without a wavefront.
//
// The phases of one direction hand their per-step state over through global memory
// (`saved`, the workspace, d_a, d_alpha).  A row stored by lane q of a quad is read
// back as a column by the other three lanes, so each phase boundary is a workgroup
// barrier (the workgroup is one wavefront; the barrier is there for its fence).
// Device code only.
#pragma once

#include "kalman_loss.h"
#include "kalman_quad.h"

namespace kvae {
namespace quad {

template <typename R, int N, int P, int K>
struct QLossAcc {
    QFilterAcc<R, N, P, K> f;   // A_K, C_K, Q, R column sums
    R m0;                       // entry q of d m0
    R P0[N];                    // column q of d P0
};

// replicated full matrix (row-major D x D) from a global pointer
template <typename R, int D>
KQ_DEV void load_full(const R* p, R* X) {
    gload<R, D * D>(p, X);
}

// column q of s * x y^T added to col (length M): col[i] += s x[i] y[q]
template <typename R, int M, int N>
KQ_DEV void outer_col_add(R s, const R* x, const R* y, R* col) {
    const R yq = s * own<N>(y);
    KV_UNROLL
    for (int i = 0; i < M; ++i) col[i] += x[i] * yq;
}

// ---------------------------------------------------------------------------
// Forward.
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
struct QLossSmoothIn {
    R m[N], mn[N], wn[K], w[K], a[P], e[N];
    QM<R, N, N> Pf, Pn;
};

template <typename R, int N, int P, int K>
KQ_DEV void loss_forward_quad(const LossFwdArgs<R>& g, int b, bool active, R* sums) {
    const long B = g.B;
    const int T = g.T;
    {
        FilterFwdArgs<R> f;
        f.T = T; f.B = g.B; f.symmetrize = g.symmetrize; f.carry_per_sequence = 0;
        f.a = g.a; f.alpha = g.alpha; f.mask = g.mask; f.A_K = g.A_K; f.C_K = g.C_K;
        f.Q = g.Q; f.Rm = g.Rm; f.m_in = g.m0; f.P_in = g.P0;
        f.m = g.m; f.Pf = g.Pf; f.mn = g.mn; f.Pn = g.Pn; f.As = nullptr; f.Cs = nullptr;
        f.status = g.status;
        filter_forward_quad<R, N, P, K>(f, b, active);
    }
    __syncthreads();

    const R on = active ? R(1) : R(0);
    int st = 0;
    R LR[P * P], LQ[N * N];
    {
        R Rm[P * P], Q[N * N];
        load_full<R, P>(g.Rm, Rm);
        load_full<R, N>(g.Q, Q);
        if (!cholesky<R, P>(Rm, LR)) st |= ST_NONPD_NOISE;
        if (!cholesky<R, N>(Q, LQ)) st |= ST_NONPD_NOISE;
    }
    const R hlR = half_logdet<R, P>(LR), hlQ = half_logdet<R, N>(LQ);
    QBases<R, N, P, K> bs;
    qload_bases<R, N, P, K>(g.A_K, g.C_K, bs);
    R s_obs = R(0), s_trans = R(0), s_post = R(0);
    R* t_obs = g.terms;
    R* t_trans = g.terms ? g.terms + (long)T * B : nullptr;
    R* t_post = g.terms ? g.terms + 2 * (long)T * B : nullptr;
    const bool writer = active && qlane() == 0;

    // z_t, obs_t and post_t from the smoothed moments of step t
    auto emit = [&](long tb, const R* ms, const QM<R, N, N>& Ps, const R* w, const R* a,
                    const R* e, R* z) {
        R F[N * N], L[N * N], r[P];
        qgather<R, N, N>(Ps, F);
        if (!cholesky<R, N>(F, L)) st |= ST_NONPD_SMOOTHED;
        vcopy<R, N>(ms, z);
        gemv<R, N, N, false, 1>(L, e, z);
        const R post = R(-0.5) * (R(N) * log_2pi<R>() + vdot<R, N>(e, e)) - half_logdet<R, N>(L);
        QM<R, P, N> C;
        qmix<R, K, P, N>(w, bs.C, C);
        vcopy<R, P>(a, r);
        mv<R, P, N, false, 2>(C, z, r);
        const R obs = gauss_logpdf_hl<R, P>(LR, hlR, r);
        s_obs += on * obs;
        s_post += on * post;
        if (t_obs && writer) {
            t_obs[tb] = obs;
            t_post[tb] = post;
        }
    };

    R ms[N], zn[N];
    QM<R, N, N> Ps;
    auto load_inputs = [&](int t, QLossSmoothIn<R, N, P, K>& in) {
        const long tb = (long)t * B + b;
        gload<R, N>(g.m + tb * N, in.m);
        qload<R, N, N>(g.Pf + tb * N * N, in.Pf);
        gload<R, N>(g.mn + tb * N, in.mn);
        qload<R, N, N>(g.Pn + tb * N * N, in.Pn);
        gload<R, K>(g.alpha + (tb + B) * K, in.wn);
        gload<R, K>(g.alpha + tb * K, in.w);
        gload<R, P>(g.a + tb * P, in.a);
        gload<R, N>(g.eps + tb * N, in.e);
    };
    auto step = [&](int t, const QLossSmoothIn<R, N, P, K>& in) {
        const long tb = (long)t * B + b;
        QM<R, N, N> An, Pi, W, J, D, T1;
        qmix<R, K, N, N>(in.wn, bs.A, An);
        if (!qspd_inverse<R, N>(in.Pn, Pi)) st |= ST_NONPD_PREDICTED;
        mm<R, N, N, N, true, false, 0>(An, Pi, W);       // An^T Pi
        mm<R, N, N, N, false, false, 0>(in.Pf, W, J);    // J = Pf An^T Pi
        R d[N];
        KV_UNROLL
        for (int i = 0; i < N; ++i) d[i] = ms[i] - in.mn[i];
        qblend<R, N, N>(R(1), Ps, R(-1), in.Pn, D);
        vcopy<R, N>(in.m, ms);
        mv<R, N, N, false, 1>(J, d, ms);
        mm<R, N, N, N, false, false, 0>(J, D, T1);
        qcopy<R, N, N>(in.Pf, Ps);
        mm<R, N, N, N, false, true, 1>(T1, J, Ps);
        if (g.symmetrize) qsymmetrize<R, N>(Ps);
        vstore<R, N>(g.ms + tb * N, ms, active);
        qstore<R, N, N>(g.Ps + tb * N * N, Ps, active);
        R z[N];
        emit(tb, ms, Ps, in.w, in.a, in.e, z);
        // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q)
        R r[N];
        vcopy<R, N>(zn, r);
        mv<R, N, N, false, 2>(An, z, r);
        const R tr = gauss_logpdf_hl<R, N>(LQ, hlQ, r);
        s_trans += on * tr;
        if (t_trans && writer) t_trans[tb + B] = tr;
        vcopy<R, N>(z, zn);
    };

    QLossSmoothIn<R, N, P, K> in0, in1;
    if (T >= 2) load_inputs(T - 2, in0);
    {
        const long tb = (long)(T - 1) * B + b;
        R w[K], a[P], e[N];
        gload<R, N>(g.m + tb * N, ms);
        qload<R, N, N>(g.Pf + tb * N * N, Ps);
        gload<R, K>(g.alpha + tb * K, w);
        gload<R, P>(g.a + tb * P, a);
        gload<R, N>(g.eps + tb * N, e);
        vstore<R, N>(g.ms + tb * N, ms, active);
        qstore<R, N, N>(g.Ps + tb * N * N, Ps, active);
        emit(tb, ms, Ps, w, a, e, zn);
    }
    for (int t = T - 2; t >= 0; t -= 2) {
        if (t >= 1) load_inputs(t - 1, in1);
        step(t, in0);
        if (t >= 1) {
            if (t >= 2) load_inputs(t - 2, in0);
            step(t - 1, in1);
        }
    }
    {
        // trans_0 = log N(z_0; m0, P0)
        R P0[N * N], L0[N * N], r[N], m0[N];
        load_full<R, N>(g.P0, P0);
        gload<R, N>(g.m0, m0);
        if (!cholesky<R, N>(P0, L0)) st |= ST_NONPD_NOISE;
        KV_UNROLL
        for (int i = 0; i < N; ++i) r[i] = zn[i] - m0[i];
        const R tr = gauss_logpdf_hl<R, N>(L0, half_logdet<R, N>(L0), r);
        s_trans += on * tr;
        if (t_trans && writer) t_trans[b] = tr;
    }
    sums[0] = s_obs;
    sums[1] = s_trans;
    sums[2] = s_post;
    if (st && active && g.status && qlane() == 0) atomicOr(g.status, st);
}

// ---------------------------------------------------------------------------
// Backward.
// ---------------------------------------------------------------------------
// what phase SA reads per step (independent of its carry, so loaded a step ahead)
template <typename R, int N, int P, int K>
struct QLossAdjIn {
    R w[K], wn[K], a[P], m[N], mn[N], ms1[N], e1[N];
    QM<R, N, N> Pf, Pn, Ps1;
};

template <typename R, int N, int P, int K>
KQ_DEV void loss_backward_quad(const LossBwdArgs<R>& g, int b, bool active,
                               QLossAcc<R, N, P, K>& acc) {
    const LossFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const R on = active ? R(1) : R(0);
    const R wo = g.w_obs * g.g_scale[0];
    const R wt = g.w_trans * g.g_scale[g.g_scale_stride];
    const R wp = g.w_post * g.g_scale[2 * g.g_scale_stride];

    R LR[P * P], LQ[N * N];
    {
        R Rm[P * P], Q[N * N];
        load_full<R, P>(f.Rm, Rm);
        load_full<R, N>(f.Q, Q);
        cholesky<R, P>(Rm, LR);
        cholesky<R, N>(Q, LQ);
    }
    QBases<R, N, P, K> bs;
    qload_bases<R, N, P, K>(f.A_K, f.C_K, bs);

    // ---- phase SA ---------------------------------------------------------------
    // sample of a step from its smoothed moments: z = ms + L eps, L = chol(Ps)
    auto sample = [&](const R* ms, const QM<R, N, N>& Ps, const R* e, R* z, R* L) {
        R F[N * N];
        qgather<R, N, N>(Ps, F);
        cholesky<R, N>(F, L);
        vcopy<R, N>(ms, z);
        gemv<R, N, N, false, 1>(L, e, z);
    };

    R z[N], L[N * N], e[N];   // step t
    {
        R ms0[N];
        QM<R, N, N> Ps0;
        gload<R, N>(f.ms + (long)b * N, ms0);
        qload<R, N, N>(f.Ps + (long)b * N * N, Ps0);
        gload<R, N>(f.eps + (long)b * N, e);
        sample(ms0, Ps0, e, z, L);
    }
    R ut[N];
    R suu_Q[N], suu_R[P];     // column q of the sums of u u^T
    vzero<R, N>(suu_Q);
    vzero<R, P>(suu_R);
    {
        // trans_0: x = z_0, mean m0, covariance P0
        R P0[N * N], L0[N * N], m0[N], P0i[N * N];
        load_full<R, N>(f.P0, P0);
        gload<R, N>(f.m0, m0);
        cholesky<R, N>(P0, L0);
        KV_UNROLL
        for (int i = 0; i < N; ++i) ut[i] = z[i] - m0[i];
        solve_lower<R, N>(L0, ut);
        solve_lower_t<R, N>(L0, ut);
        inverse_from_factor<R, N>(L0, P0i);
        acc.m0 += on * wt * own<N>(ut);
        const R uq = own<N>(ut);
        KV_UNROLL
        for (int i = 0; i < N; ++i)
            acc.P0[i] += on * wt * R(0.5) * (ut[i] * uq - own<N>(P0i + i * N));
    }

    R msb[N];
    QM<R, N, N> Psb;
    vzero<R, N>(msb);
    qzero<R, N, N>(Psb);
    R dA_row[K];
    vzero<R, K>(dA_row);

    auto load_inputs = [&](int t, QLossAdjIn<R, N, P, K>& in) {
        const long tb = (long)t * B + b;
        gload<R, K>(f.alpha + tb * K, in.w);
        gload<R, K>(f.alpha + (tb + B) * K, in.wn);
        gload<R, P>(f.a + tb * P, in.a);
        if (t + 1 < T) {
            gload<R, N>(f.m + tb * N, in.m);
            qload<R, N, N>(f.Pf + tb * N * N, in.Pf);
            gload<R, N>(f.mn + tb * N, in.mn);
            qload<R, N, N>(f.Pn + tb * N * N, in.Pn);
            gload<R, N>(f.ms + (tb + B) * N, in.ms1);
            qload<R, N, N>(f.Ps + (tb + B) * N * N, in.Ps1);
            gload<R, N>(f.eps + (tb + B) * N, in.e1);
        }
    };

    auto step = [&](int t, const QLossAdjIn<R, N, P, K>& in) {
        const long tb = (long)t * B + b;
        const bool last = t == T - 1;
        QM<R, P, N> C;
        QM<R, N, N> An;
        qmix<R, K, P, N>(in.w, bs.C, C);
        qmix<R, K, N, N>(in.wn, bs.A, An);
        R zn[N], Ln[N * N];
        if (!last) sample(in.ms1, in.Ps1, in.e1, zn, Ln);

        // -- adjoint of the terms that read z_t --
        R zb[N];
        QM<R, P, N> Cb;
        qzero<R, P, N>(Cb);
        {
            R u[P];
            vcopy<R, P>(in.a, u);
            mv<R, P, N, false, 2>(C, z, u);
            solve_lower<R, P>(LR, u);
            solve_lower_t<R, P>(LR, u);   // R^-1 (a - C z)
            R da[P];
            KV_UNROLL
            for (int i = 0; i < P; ++i) da[i] = -wo * u[i];
            vstore<R, P>(g.d_a + tb * P, da, active);
            qouter_add<R, P, N>(wo, u, z, Cb);
            mv<R, N, P, true, 0>(C, u, zb);   // C^T u
            KV_UNROLL
            for (int i = 0; i < N; ++i) zb[i] = wo * zb[i] - wt * ut[i];
            outer_col_add<R, P, P>(on, u, u, suu_R);
        }
        QM<R, N, N> Ab;
        qzero<R, N, N>(Ab);
        R un[N];
        vzero<R, N>(un);
        if (!last) {
            vcopy<R, N>(zn, un);
            mv<R, N, N, false, 2>(An, z, un);
            solve_lower<R, N>(LQ, un);
            solve_lower_t<R, N>(LQ, un);
            R s[N];
            mv<R, N, N, true, 0>(An, un, s);   // A^T u
            KV_UNROLL
            for (int i = 0; i < N; ++i) zb[i] += wt * s[i];
            qouter_add<R, N, N>(wt, un, z, Ab);
            outer_col_add<R, N, N>(on, un, un, suu_Q);
        }
        vadd<R, N>(zb, msb);
        {
            R Lb[N * N], Sadd[N * N];
            vzero<R, N * N>(Lb);
            vzero<R, N * N>(Sadd);
            outer_add<R, N, N>(R(1), zb, e, Lb);
            KV_UNROLL
            for (int i = 0; i < N; ++i) Lb[i * N + i] -= wp * recip(L[i * N + i]);
            cholesky_adjoint_add<R, N>(L, Lb, Sadd);
            QM<R, N, N> Sq;
            qscatter<R, N, N>(Sadd, Sq);
            qadd<R, N, N>(Sq, Psb);
        }
        {
            R row[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) {
                const R sC = qdot<R, P, N>(Cb, bs.C[k]);
                KV_UNROLL
                for (int i = 0; i < P; ++i) acc.f.C[k][i] += on * in.w[k] * Cb.c[i];
                row[k] = dA_row[k] + sC;
            }
            vstore<R, K>(g.d_alpha + tb * K, row, active);
        }
        if (last) {
            QM<R, N, N> z1;
            R zv[N];
            qzero<R, N, N>(z1);
            vzero<R, N>(zv);
            vstore<R, N>(g.bar_m + tb * N, msb, active);
            qstore<R, N, N>(g.bar_P + tb * N * N, Psb, active);
            vstore<R, N>(g.bar_mn + tb * N, zv, active);
            qstore<R, N, N>(g.bar_Pn + tb * N * N, z1, active);
            return;
        }
        if (t < g.burn_in) {
            vzero<R, N>(msb);
            qzero<R, N, N>(Psb);
        }

        // -- adjoint of the RTS step t (statements of smooth_backward_quad) --
        QM<R, N, N> Pi, W, J, D;
        R d[N];
        qspd_inverse<R, N>(in.Pn, Pi);
        mm<R, N, N, N, true, false, 0>(An, Pi, W);
        mm<R, N, N, N, false, false, 0>(in.Pf, W, J);
        KV_UNROLL
        for (int i = 0; i < N; ++i) d[i] = in.ms1[i] - in.mn[i];
        qblend<R, N, N>(R(1), in.Ps1, R(-1), in.Pn, D);
        if (f.symmetrize) qsymmetrize<R, N>(Psb);   // H
        vstore<R, N>(g.bar_m + tb * N, msb, active);
        QM<R, N, N> Pfb;
        qcopy<R, N, N>(Psb, Pfb);
        QM<R, N, N> Jb, T1;
        qzero<R, N, N>(Jb);
        qouter_add<R, N, N>(R(1), msb, d, Jb);
        mm<R, N, N, N, false, true, 0>(J, D, T1);         // J D^T
        mm<R, N, N, N, false, false, 1>(Psb, T1, Jb);
        mm<R, N, N, N, false, false, 0>(J, D, T1);        // J D
        mm<R, N, N, N, true, false, 1>(Psb, T1, Jb);
        R db[N];
        QM<R, N, N> Db;
        mv<R, N, N, true, 0>(J, msb, db);
        mm<R, N, N, N, true, false, 0>(J, Psb, T1);
        mm<R, N, N, N, false, false, 0>(T1, J, Db);
        mm<R, N, N, N, false, true, 1>(Jb, W, Pfb);       // Pfb += Jb W^T
        QM<R, N, N> Wb, Pib, Pnb;
        mm<R, N, N, N, true, false, 0>(in.Pf, Jb, Wb);    // Pf^T Jb
        mm<R, N, N, N, false, true, 1>(Pi, Wb, Ab);       // Ab += Pi Wb^T
        mm<R, N, N, N, false, false, 0>(An, Wb, Pib);     // Pib = An Wb
        mm<R, N, N, N, true, false, 0>(Pi, Pib, T1);
        mm<R, N, N, N, false, true, 0>(T1, Pi, Pnb);
        qblend<R, N, N>(R(-1), Pnb, R(-1), Db, Pnb);
        R mnb[N];
        KV_UNROLL
        for (int i = 0; i < N; ++i) mnb[i] = -db[i];
        qstore<R, N, N>(g.bar_P + tb * N * N, Pfb, active);
        vstore<R, N>(g.bar_mn + tb * N, mnb, active);
        qstore<R, N, N>(g.bar_Pn + tb * N * N, Pnb, active);
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            dA_row[k] = qdot<R, N, N>(Ab, bs.A[k]);
            KV_UNROLL
            for (int i = 0; i < N; ++i) acc.f.A[k][i] += on * in.wn[k] * Ab.c[i];
        }
        vcopy<R, N>(db, msb);
        qcopy<R, N, N>(Db, Psb);
        vcopy<R, N>(zn, z);
        vcopy<R, N * N>(Ln, L);
        vcopy<R, N>(in.e1, e);
        vcopy<R, N>(un, ut);
    };

    {
        QLossAdjIn<R, N, P, K> in0, in1;
        load_inputs(0, in0);
        for (int t = 0; t < T; t += 2) {
            if (t + 1 < T) load_inputs(t + 1, in1);
            step(t, in0);
            if (t + 1 < T) {
                if (t + 2 < T) load_inputs(t + 2, in0);
                step(t + 1, in1);
            }
        }
    }
    // covariance gradients of the Gaussian terms: w/2 (sum u u^T - count * Sigma^-1)
    {
        R Ri[P * P], Qi[N * N];
        inverse_from_factor<R, P>(LR, Ri);
        inverse_from_factor<R, N>(LQ, Qi);
        KV_UNROLL
        for (int i = 0; i < P; ++i)
            acc.f.Rr[i] += wo * R(0.5) * (suu_R[i] - on * R(T) * own<P>(Ri + i * P));
        KV_UNROLL
        for (int i = 0; i < N; ++i)
            acc.f.Q[i] += wt * R(0.5) * (suu_Q[i] - on * R(T - 1) * own<N>(Qi + i * N));
    }
    __syncthreads();

    // ---- phase FA ---------------------------------------------------------------
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
    R m0b[N];
    QM<R, N, N> P0b;
    filter_backward_quad_core<R, N, P, K, true>(fb, b, active, acc.f, m0b, P0b);
    acc.m0 += on * own<N>(m0b);
    KV_UNROLL
    for (int i = 0; i < N; ++i) acc.P0[i] += on * P0b.c[i];
}

// row `row` of the partial-sum matrix (layout LossParamGrads) from the column sums
template <typename R, int N, int P, int K>
KQ_DEV void write_loss_partial(const QLossAcc<R, N, P, K>& acc, R* partial, long row) {
    using PG = LossParamGrads<N, P, K>;
    const int lane = threadIdx.x & 63, q = lane & 3;
    R* out = partial + row * PG::LEN;
    write_filter_partial_to<R, N, P, K>(acc.f, out);
    {
        const R v = sum_over_quads(acc.m0);
        if (lane < 4 && q < N) out[PG::OFF_M0 + q] = v;
    }
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        const R v = sum_over_quads(acc.P0[i]);
        if (lane < 4 && q < N) out[PG::OFF_P0 + i * N + q] = v;
    }
}

}  // namespace quad
}  // namespace kvae
