// Four lanes per sequence ("quad") form of the small-state filter and smoother.
//
// The lane-per-sequence kernels of kalman_core.h issue ~3000 instructions per
// step from a single wavefront; a batch of a few thousand sequences is only a
// few dozen wavefronts, so most SIMDs of the chip idle and the kernels are bound
// by one wavefront's issue rate.  Here the four lanes of a quad share one
// sequence and a wavefront holds 16 sequences.  Every matrix (n <= 4, p <= 4, so
// at most 4 x 4) is kept distributed over the quad in BOTH orientations:
//   c[i] = X[i][q]   (lane q holds column q)     t[j] = X[q][j]   (lane q holds row q)
// with zeros where q is outside the matrix.  In that layout the float product
// X^T Y of two column-distributed operands is exactly what the 16-block MFMA
// v_mfma_f32_4x4x1_16b_f32 computes (block = quad = sequence), 4 instructions per
// 4x4x4 product, and every product op(A) op(B) of the recursion is such a product
// of the right orientations; its transpose is the same instruction with the
// operands swapped.  Orientations that nothing reads are dead code and the
// compiler drops them.  double has no such MFMA: there the same product runs on
// DPP quad broadcasts.  Vectors are replicated in the four lanes; the tiny
// factorisations (inverse of the p x p innovation covariance, Cholesky of an
// n x n covariance and its adjoint) gather the matrix into every lane and run the
// lane code of kalman_core.h redundantly.
//
// Math, adjoint derivations and the reference line numbers are those of
// kalman_core.h (filter_forward_sequence etc.); this file only re-expresses them
// on the distributed layout.  Device code only.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kalman_core.h"

namespace kvae {
namespace quad {

#define KQ_DEV __device__ __forceinline__

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
KQ_DEV void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

KQ_DEV int qlane() { return threadIdx.x & 3; }

template <int CTRL>
KQ_DEV float dpp(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
KQ_DEV double dpp(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

// value of x in lane K of this quad
template <int K, typename R>
KQ_DEV R bc(R x) {
    return dpp<K * 0x55>(x);
}
template <typename R>
KQ_DEV R xor1(R x) { return dpp<0xB1>(x); }   // quad_perm [1,0,3,2]
template <typename R>
KQ_DEV R xor2(R x) { return dpp<0x4E>(x); }   // quad_perm [2,3,0,1]

// sum over the quad, result in all four lanes
template <typename R>
KQ_DEV R qsum(R x) {
    x += xor1(x);
    x += xor2(x);
    return x;
}

// x[q] for this lane's q (0 when q >= N); x is replicated
template <int N, typename R>
KQ_DEV R own(const R* x) {
    const int q = qlane();
    R r = R(0);
    KV_UNROLL
    for (int j = 0; j < N; ++j) r = (q == j) ? x[j] : r;
    return r;
}

typedef float f32x4_q __attribute__((ext_vector_type(4)));

// A matrix of M rows and N columns (both <= 4), both orientations.
template <typename R, int M, int N>
struct QM {
    R c[M];   // column q of X, zero when q >= N
    R t[N];   // row q of X,    zero when q >= M
};

// C (M entries: column q of the M x N result) {=,+=,-=} X^T Y, X = column q of a
// Kd x M matrix, Y = column q of a Kd x N matrix.
template <typename R, int M, int Kd, int N, int MODE>
KQ_DEV void tn(const R* X, const R* Y, R* C) {
    if constexpr (std::is_same<R, float>::value) {
        f32x4_q acc;
        KV_UNROLL
        for (int i = 0; i < 4; ++i) acc[i] = (MODE != 0 && i < M) ? C[i] : 0.f;
        KV_UNROLL
        for (int k = 0; k < Kd; ++k)
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(MODE == 2 ? -X[k] : X[k], Y[k], acc, 0, 0, 0);
        KV_UNROLL
        for (int i = 0; i < M; ++i) C[i] = acc[i];
    } else {
        R acc[M];
        KV_UNROLL
        for (int i = 0; i < M; ++i) acc[i] = R(0);
        sfor<0, M>([&](auto i) {
            KV_UNROLL
            for (int k = 0; k < Kd; ++k) acc[i.value] += bc<i.value>(X[k]) * Y[k];
        });
        KV_UNROLL
        for (int i = 0; i < M; ++i) {
            if (MODE == 0) C[i] = acc[i];
            if (MODE == 1) C[i] += acc[i];
            if (MODE == 2) C[i] -= acc[i];
        }
    }
}

// Z (M x N) {=,+=,-=} op(A) op(B), flags and MODE as kalman_core.h's gemm:
// A is stored M x Kd (Kd x M when TA), B is stored Kd x N (N x Kd when TB).
template <typename R, int M, int Kd, int N, bool TA, bool TB, int MODE, typename MA, typename MB>
KQ_DEV void mm(const MA& A, const MB& Bm, QM<R, M, N>& Z) {
    const R* X = TA ? A.c : A.t;     // column q of op(A)^T  (Kd x M)
    const R* Y = TB ? Bm.t : Bm.c;   // column q of op(B)    (Kd x N)
    tn<R, M, Kd, N, MODE>(X, Y, Z.c);
    tn<R, N, Kd, M, MODE>(Y, X, Z.t);
}

// y (M, replicated) {=,+=,-=} op(A) x ; A stored M x N (N x M when TA)
template <typename R, int M, int N, bool TA, int MODE, typename MA>
KQ_DEV void mv(const MA& A, const R* x, R* y) {
    const R* col = TA ? A.t : A.c;   // column q of op(A)
    const R xq = own<N>(x);
    KV_UNROLL
    for (int i = 0; i < M; ++i) {
        const R v = qsum(col[i] * xq);
        if (MODE == 0) y[i] = v;
        if (MODE == 1) y[i] += v;
        if (MODE == 2) y[i] -= v;
    }
}

// C (M x N) += s * x y^T ; x (M), y (N) replicated
template <typename R, int M, int N>
KQ_DEV void qouter_add(R s, const R* x, const R* y, QM<R, M, N>& C) {
    const R yq = s * own<N>(y), xq = s * own<M>(x);
    KV_UNROLL
    for (int i = 0; i < M; ++i) C.c[i] += x[i] * yq;
    KV_UNROLL
    for (int j = 0; j < N; ++j) C.t[j] += y[j] * xq;
}

template <typename R, int N>
KQ_DEV void qsymmetrize(QM<R, N, N>& X) {
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        X.c[i] = (X.c[i] + X.t[i]) / R(2);
        X.t[i] = X.c[i];
    }
}

template <typename R, int M, int N>
KQ_DEV void qzero(QM<R, M, N>& X) {
    vzero<R, M>(X.c);
    vzero<R, N>(X.t);
}
template <typename R, int M, int N>
KQ_DEV void qcopy(const QM<R, M, N>& S, QM<R, M, N>& D) {
    vcopy<R, M>(S.c, D.c);
    vcopy<R, N>(S.t, D.t);
}
template <typename R, int M, int N>
KQ_DEV void qadd(const QM<R, M, N>& S, QM<R, M, N>& D) {
    vadd<R, M>(S.c, D.c);
    vadd<R, N>(S.t, D.t);
}
// D = a * S
template <typename R, int M, int N>
KQ_DEV void qscale(R a, const QM<R, M, N>& S, QM<R, M, N>& D) {
    KV_UNROLL
    for (int i = 0; i < M; ++i) D.c[i] = a * S.c[i];
    KV_UNROLL
    for (int j = 0; j < N; ++j) D.t[j] = a * S.t[j];
}
// D = a * S + b * U
template <typename R, int M, int N>
KQ_DEV void qblend(R a, const QM<R, M, N>& S, R b, const QM<R, M, N>& U, QM<R, M, N>& D) {
    KV_UNROLL
    for (int i = 0; i < M; ++i) D.c[i] = a * S.c[i] + b * U.c[i];
    KV_UNROLL
    for (int j = 0; j < N; ++j) D.t[j] = a * S.t[j] + b * U.t[j];
}

// row-major M x N matrix in global memory -> both orientations
template <typename R, int M, int N>
KQ_DEV void qload(const R* p, QM<R, M, N>& X) {
    const int q = qlane();
    const int qc = q < N ? q : 0, qr = q < M ? q : 0;
    KV_UNROLL
    for (int i = 0; i < M; ++i) {
        const R v = p[i * N + qc];
        X.c[i] = q < N ? v : R(0);
    }
    KV_UNROLL
    for (int j = 0; j < N; ++j) {
        const R v = p[qr * N + j];
        X.t[j] = q < M ? v : R(0);
    }
}
template <typename R, int M, int N>
KQ_DEV void qload_or_zero(const R* base, long off, QM<R, M, N>& X) {
    if (base) {
        qload<R, M, N>(base + off, X);
    } else {
        qzero<R, M, N>(X);
    }
}
// lane q stores row q (contiguous)
template <typename R, int M, int N>
KQ_DEV void qstore(R* p, const QM<R, M, N>& X, bool active) {
    const int q = qlane();
    if (active && q < M) {
        KV_UNROLL
        for (int j = 0; j < N; ++j) p[q * N + j] = X.t[j];
    }
}
// replicated vector: every lane loads it (gload); lane 0 of the quad stores it
template <typename R, int Len>
KQ_DEV void vstore(R* p, const R* x, bool active) {
    if (active && qlane() == 0) gstore<R, Len>(p, x);
}

// full (M x N row-major, replicated in the four lanes) <-> distributed
template <typename R, int M, int N>
KQ_DEV void qgather(const QM<R, M, N>& X, R* full) {
    sfor<0, N>([&](auto j) {
        KV_UNROLL
        for (int i = 0; i < M; ++i) full[i * N + j.value] = bc<j.value>(X.c[i]);
    });
}
template <typename R, int M, int N>
KQ_DEV void qscatter(const R* full, QM<R, M, N>& X) {
    const int q = qlane();
    KV_UNROLL
    for (int i = 0; i < M; ++i) X.c[i] = own<N>(full + i * N);
    KV_UNROLL
    for (int j = 0; j < N; ++j) {
        R r = R(0);
        KV_UNROLL
        for (int i = 0; i < M; ++i) r = (q == i) ? full[i * N + j] : r;
        X.t[j] = r;
    }
}

// sum_k w[k] bases[k]
template <typename R, int K, int M, int N>
KQ_DEV void qmix(const R* w, const QM<R, M, N>* bases, QM<R, M, N>& out) {
    KV_UNROLL
    for (int i = 0; i < M; ++i) {
        R acc = R(0);
        KV_UNROLL
        for (int k = 0; k < K; ++k) acc += w[k] * bases[k].c[i];
        out.c[i] = acc;
    }
    KV_UNROLL
    for (int j = 0; j < N; ++j) {
        R acc = R(0);
        KV_UNROLL
        for (int k = 0; k < K; ++k) acc += w[k] * bases[k].t[j];
        out.t[j] = acc;
    }
}

// <X, Y> = sum of the element-wise product (replicated result)
template <typename R, int M, int N>
KQ_DEV R qdot(const QM<R, M, N>& X, const QM<R, M, N>& Y) {
    R s = R(0);
    KV_UNROLL
    for (int i = 0; i < M; ++i) s += X.c[i] * Y.c[i];
    return qsum(s);
}

// S^-1 of an SPD matrix; optionally the replicated Cholesky factor.
template <typename R, int N>
KQ_DEV bool qspd_inverse(const QM<R, N, N>& S, QM<R, N, N>& Si, R* L_full = nullptr) {
    R F[N * N], Fi[N * N], L[N * N];
    qgather<R, N, N>(S, F);
    const bool ok = spd_inverse<R, N>(F, Fi, L, nullptr);
    qscatter<R, N, N>(Fi, Si);
    if (L_full) vcopy<R, N * N>(L, L_full);
    return ok;
}

// ---------------------------------------------------------------------------
// Filter, forward.  Same statements as filter_forward_sequence.
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
struct QUpdateTerms {
    QM<R, P, N> M;    // C Pp
    QM<R, P, P> Si;   // (C Pp C^T + R)^-1
    QM<R, N, P> F;    // Pp C^T
    QM<R, N, P> G;    // gain F Si
    R r[P];           // innovation a - C mp (replicated)
};

template <typename R, int N, int P>
KQ_DEV bool qupdate_terms(const QM<R, P, N>& C, const R* mp, const QM<R, N, N>& Pp,
                          const QM<R, P, P>& Rm, const R* a, QUpdateTerms<R, N, P>& u) {
    QM<R, P, P> S;
    mm<R, P, N, N, false, false, 0>(C, Pp, u.M);
    qcopy<R, P, P>(Rm, S);
    mm<R, P, N, P, false, true, 1>(u.M, C, S);   // S = M C^T + R
    const bool ok = qspd_inverse<R, P>(S, u.Si);
    mm<R, N, N, P, false, true, 0>(Pp, C, u.F);
    mm<R, N, P, P, false, false, 0>(u.F, u.Si, u.G);
    vcopy<R, P>(a, u.r);
    mv<R, P, N, false, 2>(C, mp, u.r);
    return ok;
}

template <typename R, int N, int P, int K>
struct QBases {
    QM<R, N, N> A[K];
    QM<R, P, N> C[K];
};
template <typename R, int N, int P, int K>
KQ_DEV void qload_bases(const R* A_K, const R* C_K, QBases<R, N, P, K>& bs) {
    KV_UNROLL
    for (int k = 0; k < K; ++k) {
        qload<R, N, N>(A_K + k * N * N, bs.A[k]);
        qload<R, P, N>(C_K + k * P * N, bs.C[k]);
    }
}

template <typename R, int N, int P, int K>
KQ_DEV void filter_forward_quad(const FilterFwdArgs<R>& g, int b, bool active) {
    const long B = g.B;
    QM<R, P, P> Rm;
    QM<R, N, N> Q;
    QBases<R, N, P, K> bs;
    qload<R, P, P>(g.Rm, Rm);
    qload<R, N, N>(g.Q, Q);
    qload_bases<R, N, P, K>(g.A_K, g.C_K, bs);
    R mp[N];
    QM<R, N, N> Pp;
    if (g.carry_per_sequence) {
        gload<R, N>(g.m_in + (long)b * N, mp);
        qload<R, N, N>(g.P_in + (long)b * N * N, Pp);
    } else {
        gload<R, N>(g.m_in, mp);
        qload<R, N, N>(g.P_in, Pp);
    }
    R w[K], wn[K];
    gload<R, K>(g.alpha + (long)b * K, w);
    if (g.As) {
        QM<R, N, N> A0;
        qmix<R, K, N, N>(w, bs.A, A0);
        qstore<R, N, N>(g.As + (long)b * N * N, A0, active);
    }
    int st = 0;
    // the inputs of step t+1 do not depend on the recursion: they are loaded while
    // step t computes, so their memory latency is off the serial chain
    R wn_next[K], a_next[P], mu_next = R(1);
    gload<R, K>(g.alpha + (B + b) * K, wn_next);
    gload<R, P>(g.a + (long)b * P, a_next);
    if (g.mask) mu_next = g.mask[b];
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        QM<R, P, N> C;
        QM<R, N, N> An;
        R a[P];
        vcopy<R, K>(wn_next, wn);
        vcopy<R, P>(a_next, a);
        const R mu = mu_next;
        if (t + 1 < g.T) {
            gload<R, K>(g.alpha + (tb + 2 * B) * K, wn_next);
            gload<R, P>(g.a + (tb + B) * P, a_next);
            if (g.mask) mu_next = g.mask[tb + B];
        }
        qmix<R, K, P, N>(w, bs.C, C);
        qmix<R, K, N, N>(wn, bs.A, An);
        if (g.Cs) qstore<R, P, N>(g.Cs + tb * P * N, C, active);
        if (g.As) qstore<R, N, N>(g.As + (tb + B) * N * N, An, active);

        QUpdateTerms<R, N, P> u;
        if (!qupdate_terms<R, N, P>(C, mp, Pp, Rm, a, u)) st |= ST_NONPD_INNOVATION;

        R m[N];
        QM<R, N, N> Pf;
        vcopy<R, N>(mp, m);
        mv<R, N, P, false, 1>(u.G, u.r, m);                    // m_obs = mp + G r
        qcopy<R, N, N>(Pp, Pf);
        mm<R, N, P, N, false, false, 2>(u.G, u.M, Pf);         // P_obs = Pp - G (C Pp)
        if (g.mask) {
            const R nu = R(1) - mu;
            KV_UNROLL
            for (int i = 0; i < N; ++i) m[i] = mu * m[i] + nu * mp[i];
            qblend<R, N, N>(mu, Pf, nu, Pp, Pf);
        }
        if (g.symmetrize) qsymmetrize<R, N>(Pf);

        QM<R, N, N> T1;
        mv<R, N, N, false, 0>(An, m, mp);                      // m_{t+1|t}
        mm<R, N, N, N, false, false, 0>(An, Pf, T1);           // An P
        qcopy<R, N, N>(Q, Pp);
        mm<R, N, N, N, false, true, 1>(T1, An, Pp);            // An P An^T + Q
        if (g.symmetrize) qsymmetrize<R, N>(Pp);

        vstore<R, N>(g.m + tb * N, m, active);
        qstore<R, N, N>(g.Pf + tb * N * N, Pf, active);
        vstore<R, N>(g.mn + tb * N, mp, active);
        qstore<R, N, N>(g.Pn + tb * N * N, Pp, active);
        vcopy<R, K>(wn, w);
    }
    if (g.Cs) {
        QM<R, P, N> C;
        qmix<R, K, P, N>(w, bs.C, C);
        qstore<R, P, N>(g.Cs + ((long)g.T * B + b) * P * N, C, active);
    }
    if (st && active && g.status && qlane() == 0) atomicOr(g.status, st);
}

// ---------------------------------------------------------------------------
// Filter, backward.  Same statements as filter_backward_sequence; the running
// parameter-gradient sums hold column q of each gradient matrix.
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
struct QFilterAcc {
    R A[K][N];   // column q of d A_K[k]
    R C[K][P];   // column q of d C_K[k]
    R Q[N];
    R Rr[P];
};

// what one backward step reads from global memory
template <typename R, int N, int P, int K>
struct QFilterBwdIn {
    R w[K], wn[K], a[P], mp[N], m[N], mb[N], gmn[N], mu;
    R da_part[P], dal_part[K];   // ACCUM: what d_a[t], d_alpha[t+1] already hold
    QM<R, N, N> Pp, Pf, Pb, gPn, Ab;
    QM<R, P, N> Cb;
};

// ACCUM: d_a and rows 0..T-1 of d_alpha already hold a first part of the gradient
// (kalman_core.h: filter_backward_core) and this pass adds to it.  The gradient w.r.t.
// the initial (m_in, P_in) is returned in m_in_bar (replicated) / P_in_bar.
template <typename R, int N, int P, int K, bool ACCUM>
KQ_DEV void filter_backward_quad_core(const FilterBwdArgs<R>& g, int b, bool active,
                                      QFilterAcc<R, N, P, K>& acc, R* m_in_bar,
                                      QM<R, N, N>& P_in_bar) {
    const FilterFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const R on = active ? R(1) : R(0);   // sequences past the end add nothing to the sums
    QM<R, P, P> Rm;
    QBases<R, N, P, K> bs;
    qload<R, P, P>(f.Rm, Rm);
    qload_bases<R, N, P, K>(f.A_K, f.C_K, bs);

    R mpb[N];
    QM<R, N, N> Ppb;
    vzero<R, N>(mpb);
    qzero<R, N, N>(Ppb);
    R dwn[K];
    {
        QM<R, P, N> gC;
        R wT[K];
        qload_or_zero<R, P, N>(g.g_Cs, ((long)T * B + b) * P * N, gC);
        gload<R, K>(f.alpha + ((long)T * B + b) * K, wT);
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            dwn[k] = qdot<R, P, N>(gC, bs.C[k]);
            KV_UNROLL
            for (int i = 0; i < P; ++i) acc.C[k][i] += on * wT[k] * gC.c[i];
        }
    }

    // Everything a step reads from memory is independent of the recursion, so the
    // inputs of step t-1 are loaded while step t computes (two buffers used in turn).
    auto load_inputs = [&](int t, QFilterBwdIn<R, N, P, K>& in) {
        const long tb = (long)t * B + b;
        gload<R, K>(f.alpha + tb * K, in.w);
        gload<R, K>(f.alpha + (tb + B) * K, in.wn);
        gload<R, P>(f.a + tb * P, in.a);
        if (t > 0) {
            gload<R, N>(f.mn + (tb - B) * N, in.mp);
            qload<R, N, N>(f.Pn + (tb - B) * N * N, in.Pp);
        } else if (f.carry_per_sequence) {
            gload<R, N>(f.m_in + (long)b * N, in.mp);
            qload<R, N, N>(f.P_in + (long)b * N * N, in.Pp);
        } else {
            gload<R, N>(f.m_in, in.mp);
            qload<R, N, N>(f.P_in, in.Pp);
        }
        gload<R, N>(f.m + tb * N, in.m);
        qload<R, N, N>(f.Pf + tb * N * N, in.Pf);
        gload_or_zero<R, N>(g.g_m, tb * N, in.mb);
        qload_or_zero<R, N, N>(g.g_P, tb * N * N, in.Pb);
        gload_or_zero<R, N>(g.g_mn, tb * N, in.gmn);
        qload_or_zero<R, N, N>(g.g_Pn, tb * N * N, in.gPn);
        qload_or_zero<R, N, N>(g.g_As, (tb + B) * N * N, in.Ab);
        qload_or_zero<R, P, N>(g.g_Cs, tb * P * N, in.Cb);
        in.mu = f.mask ? f.mask[tb] : R(1);
        if (ACCUM) {
            gload<R, P>(g.d_a + tb * P, in.da_part);
            if (t + 1 < T) {
                gload<R, K>(g.d_alpha + (tb + B) * K, in.dal_part);
            } else {
                vzero<R, K>(in.dal_part);
            }
        }
    };

    auto step = [&](int t, const QFilterBwdIn<R, N, P, K>& in) {
        const long tb = (long)t * B + b;
        R w[K], wn[K], a[P];
        QM<R, P, N> C;
        QM<R, N, N> An;
        vcopy<R, K>(in.w, w);
        vcopy<R, K>(in.wn, wn);
        vcopy<R, P>(in.a, a);
        qmix<R, K, P, N>(w, bs.C, C);
        qmix<R, K, N, N>(wn, bs.A, An);

        R mp[N], m[N];
        QM<R, N, N> Pp, Pf;
        vcopy<R, N>(in.mp, mp);
        vcopy<R, N>(in.m, m);
        qcopy<R, N, N>(in.Pp, Pp);
        qcopy<R, N, N>(in.Pf, Pf);

        // 1. gradients of the step's outputs
        R mb[N];
        QM<R, N, N> Pb;
        vcopy<R, N>(in.mb, mb);
        qcopy<R, N, N>(in.Pb, Pb);
        vadd<R, N>(in.gmn, mpb);
        qadd<R, N, N>(in.gPn, Ppb);
        if (t < g.burn_in) {
            vzero<R, N>(mb);
            qzero<R, N, N>(Pb);
            vzero<R, N>(mpb);
            qzero<R, N, N>(Ppb);
        }

        // 2. time update adjoint
        QM<R, N, N> Ab;
        qcopy<R, N, N>(in.Ab, Ab);
        {
            if (f.symmetrize) qsymmetrize<R, N>(Ppb);   // H
            QM<R, N, N> T1, T2;
            qouter_add<R, N, N>(R(1), mpb, m, Ab);
            mm<R, N, N, N, false, true, 0>(An, Pf, T1);    // An Pf^T
            mm<R, N, N, N, false, false, 1>(Ppb, T1, Ab);
            mm<R, N, N, N, false, false, 0>(An, Pf, T1);   // An Pf
            mm<R, N, N, N, true, false, 1>(Ppb, T1, Ab);
            mv<R, N, N, true, 1>(An, mpb, mb);
            mm<R, N, N, N, true, false, 0>(An, Ppb, T2);   // An^T H
            mm<R, N, N, N, false, false, 1>(T2, An, Pb);
            KV_UNROLL
            for (int i = 0; i < N; ++i) acc.Q[i] += on * Ppb.c[i];
        }

        // 3. measurement update adjoint
        QUpdateTerms<R, N, P> u;
        qupdate_terms<R, N, P>(C, mp, Pp, Rm, a, u);
        const R mu = in.mu;
        const R nu = R(1) - mu;
        if (f.symmetrize) qsymmetrize<R, N>(Pb);   // H
        QM<R, P, N> Cb;
        qcopy<R, P, N>(in.Cb, Cb);
        R mob[N];
        QM<R, N, N> Pob;
        KV_UNROLL
        for (int i = 0; i < N; ++i) {
            mob[i] = mu * mb[i];
            mpb[i] = nu * mb[i] + mob[i];
        }
        qscale<R, N, N>(mu, Pb, Pob);
        qblend<R, N, N>(nu, Pb, R(1), Pob, Ppb);
        QM<R, N, P> Gb;
        qzero<R, N, P>(Gb);
        qouter_add<R, N, P>(R(1), mob, u.r, Gb);
        R rb[P];
        mv<R, P, N, true, 0>(u.G, mob, rb);               // G^T mob
        if (ACCUM) {
            R da_t[P];
            KV_UNROLL
            for (int i = 0; i < P; ++i) da_t[i] = in.da_part[i] + rb[i];
            vstore<R, P>(g.d_a + tb * P, da_t, active);
        } else {
            vstore<R, P>(g.d_a + tb * P, rb, active);
        }
        qouter_add<R, P, N>(R(-1), rb, mp, Cb);
        mv<R, N, P, true, 2>(C, rb, mpb);                 // mpb -= C^T rb
        mm<R, N, N, P, false, true, 2>(Pob, u.M, Gb);     // Gb -= Pob M^T
        QM<R, P, N> Mb;
        mm<R, P, N, N, true, false, 0>(u.G, Pob, Mb);     // G^T Pob (sign below)
        mm<R, P, N, N, false, true, 2>(Mb, Pp, Cb);
        mm<R, N, P, N, true, false, 2>(C, Mb, Ppb);
        QM<R, N, P> Fb;
        QM<R, P, P> Sib, Sb, T3;
        mm<R, N, P, P, false, true, 0>(Gb, u.Si, Fb);     // Gb Si^T
        mm<R, P, N, P, true, false, 0>(u.F, Gb, Sib);     // F^T Gb
        mm<R, P, P, P, true, false, 0>(u.Si, Sib, T3);
        mm<R, P, P, P, false, true, 0>(T3, u.Si, Sb);
        qscale<R, P, P>(R(-1), Sb, Sb);
        mm<R, N, P, N, false, false, 1>(Fb, C, Ppb);      // Ppb += Fb C
        mm<R, P, N, N, true, false, 1>(Fb, Pp, Cb);       // Cb += Fb^T Pp
        {
            QM<R, P, N> CPt, T4;
            mm<R, P, N, N, false, true, 0>(C, Pp, CPt);   // C Pp^T
            mm<R, P, P, N, false, false, 1>(Sb, CPt, Cb);
            mm<R, P, P, N, true, false, 1>(Sb, u.M, Cb);
            mm<R, P, P, N, false, false, 0>(Sb, C, T4);
            mm<R, N, P, N, true, false, 1>(C, T4, Ppb);   // C^T Sb C
            KV_UNROLL
            for (int i = 0; i < P; ++i) acc.Rr[i] += on * Sb.c[i];
        }

        // 4. mixing adjoint
        {
            R da[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) {
                const R sA = qdot<R, N, N>(Ab, bs.A[k]);
                const R sC = qdot<R, P, N>(Cb, bs.C[k]);
                KV_UNROLL
                for (int i = 0; i < N; ++i) acc.A[k][i] += on * wn[k] * Ab.c[i];
                KV_UNROLL
                for (int i = 0; i < P; ++i) acc.C[k][i] += on * w[k] * Cb.c[i];
                da[k] = dwn[k] + sA;
                if (ACCUM) da[k] += in.dal_part[k];
                dwn[k] = sC;
            }
            vstore<R, K>(g.d_alpha + (tb + B) * K, da, active);
        }
    };

    QFilterBwdIn<R, N, P, K> in0, in1;
    load_inputs(T - 1, in0);
    for (int t = T - 1; t >= 0; t -= 2) {
        if (t >= 1) load_inputs(t - 1, in1);
        step(t, in0);
        if (t >= 1) {
            if (t >= 2) load_inputs(t - 2, in0);
            step(t - 1, in1);
        }
    }
    {
        QM<R, N, N> gA;
        R w0[K], da[K];
        qload_or_zero<R, N, N>(g.g_As, (long)b * N * N, gA);
        gload<R, K>(f.alpha + (long)b * K, w0);
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            const R s = qdot<R, N, N>(gA, bs.A[k]);
            KV_UNROLL
            for (int i = 0; i < N; ++i) acc.A[k][i] += on * w0[k] * gA.c[i];
            da[k] = dwn[k] + s;
            if (ACCUM) da[k] += g.d_alpha[(long)b * K + k];
        }
        vstore<R, K>(g.d_alpha + (long)b * K, da, active);
    }
    vcopy<R, N>(mpb, m_in_bar);
    qcopy<R, N, N>(Ppb, P_in_bar);
}

template <typename R, int N, int P, int K>
KQ_DEV void filter_backward_quad(const FilterBwdArgs<R>& g, int b, bool active,
                                 QFilterAcc<R, N, P, K>& acc) {
    R mpb[N];
    QM<R, N, N> Ppb;
    filter_backward_quad_core<R, N, P, K, false>(g, b, active, acc, mpb, Ppb);
    vstore<R, N>(g.d_m_in + (long)b * N, mpb, active);
    qstore<R, N, N>(g.d_P_in + (long)b * N * N, Ppb, active);
}

// ---------------------------------------------------------------------------
// Smoother.  Same statements as smooth_forward_sequence / smooth_backward_sequence.
// The Cholesky factor of the smoothed covariance (sampling) and its adjoint run
// replicated in the four lanes on the gathered matrix.
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
KQ_DEV bool qsmooth_emit(const SmoothFwdArgs<R>& g, long tb, const R* mean, const QM<R, N, N>& cov,
                         const QM<R, P, N>& C, const R* LR, const R* es, const R* eo,
                         bool active) {
    bool ok = true;
    R z[N], ar[P];
    vcopy<R, N>(mean, z);
    if (g.eps_s) {
        R F[N * N], L[N * N];
        qgather<R, N, N>(cov, F);
        ok = cholesky<R, N>(F, L);
        gemv<R, N, N, false, 1>(L, es, z);
    }
    mv<R, P, N, false, 0>(C, z, ar);
    if (g.eps_o) gemv<R, P, P, false, 1>(LR, eo, ar);
    vstore<R, N>(g.zs + tb * N, z, active);
    vstore<R, P>(g.ares + tb * P, ar, active);
    return ok;
}

template <typename R, int N, int P>
struct QSmoothIn {
    R m[N], mn[N], es[N], eo[P];
    QM<R, N, N> Pf, Pn, An;
    QM<R, P, N> C;
};

template <typename R, int N, int P>
KQ_DEV void smooth_forward_quad(const SmoothFwdArgs<R>& g, int b, bool active) {
    const long B = g.B;
    const int T = g.T;
    R LR[P * P];
    if (g.eps_o) {
        gload<R, P * P>(g.LR, LR);
    } else {
        vzero<R, P * P>(LR);
    }
    int st = 0;
    R ms[N];
    QM<R, N, N> Ps;
    // the inputs of a step do not depend on the recursion: step t-1's are loaded
    // while step t computes (two buffers used in turn)
    auto load_noise = [&](long tb, R* es, R* eo) {
        gload_or_zero<R, N>(g.eps_s, tb * N, es);
        gload_or_zero<R, P>(g.eps_o, tb * P, eo);
    };
    auto load_inputs = [&](int t, QSmoothIn<R, N, P>& in) {
        const long tb = (long)t * B + b;
        gload<R, N>(g.m + tb * N, in.m);
        qload<R, N, N>(g.Pf + tb * N * N, in.Pf);
        gload<R, N>(g.mn + tb * N, in.mn);
        qload<R, N, N>(g.Pn + tb * N * N, in.Pn);
        qload<R, N, N>(g.As + (tb + B) * N * N, in.An);
        qload<R, P, N>(g.Cs + tb * P * N, in.C);
        load_noise(tb, in.es, in.eo);
    };
    auto step = [&](int t, const QSmoothIn<R, N, P>& in) {
        const long tb = (long)t * B + b;
        QM<R, N, N> Pi, W, J, D, T1;
        if (!qspd_inverse<R, N>(in.Pn, Pi)) st |= ST_NONPD_PREDICTED;
        mm<R, N, N, N, true, false, 0>(in.An, Pi, W);    // An^T Pi
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
        if (!qsmooth_emit<R, N, P>(g, tb, ms, Ps, in.C, LR, in.es, in.eo, active))
            st |= ST_NONPD_SMOOTHED;
    };
    QSmoothIn<R, N, P> in0, in1;
    if (T >= 2) load_inputs(T - 2, in0);
    {
        const long tb = (long)(T - 1) * B + b;
        QM<R, P, N> C;
        R es[N], eo[P];
        gload<R, N>(g.m + tb * N, ms);
        qload<R, N, N>(g.Pf + tb * N * N, Ps);
        qload<R, P, N>(g.Cs + (tb + B) * P * N, C);   // C_T, the extra entry
        load_noise(tb, es, eo);
        vstore<R, N>(g.ms + tb * N, ms, active);
        qstore<R, N, N>(g.Ps + tb * N * N, Ps, active);
        if (!qsmooth_emit<R, N, P>(g, tb, ms, Ps, C, LR, es, eo, active)) st |= ST_NONPD_SMOOTHED;
    }
    for (int t = T - 2; t >= 0; t -= 2) {
        if (t >= 1) load_inputs(t - 1, in1);
        step(t, in0);
        if (t >= 1) {
            if (t >= 2) load_inputs(t - 2, in0);
            step(t - 1, in1);
        }
    }
    if (st && active && g.status && qlane() == 0) atomicOr(g.status, st);
}

// acc_LR: P*P running sum of the gradient of LR, replicated in the quad (or null)
template <typename R, int N, int P>
KQ_DEV void qsmooth_emit_adjoint(const SmoothBwdArgs<R>& g, const R* mean, const QM<R, N, N>& cov,
                                 const QM<R, P, N>& C, const R* zb_in, const R* ab, const R* es,
                                 const R* eo, R* msb, QM<R, N, N>& Psb, QM<R, P, N>& Cb,
                                 R* acc_LR, R on) {
    const SmoothFwdArgs<R>& f = g.f;
    R zb[N];
    vcopy<R, N>(zb_in, zb);
    qzero<R, P, N>(Cb);
    R L[N * N];
    if (f.eps_s && (g.g_ares || g.g_zs)) {
        R F[N * N];
        qgather<R, N, N>(cov, F);
        cholesky<R, N>(F, L);
    }
    if (g.g_ares) {
        R z[N];
        vcopy<R, N>(mean, z);
        if (f.eps_s) gemv<R, N, N, false, 1>(L, es, z);
        qouter_add<R, P, N>(R(1), ab, z, Cb);
        mv<R, N, P, true, 1>(C, ab, zb);              // zb += C^T ab
        if (f.eps_o && acc_LR) outer_add<R, P, P>(on, ab, eo, acc_LR);
    }
    if (g.g_ares || g.g_zs) {
        vadd<R, N>(zb, msb);
        if (f.eps_s) {
            R Lb[N * N], Sadd[N * N];
            vzero<R, N * N>(Lb);
            vzero<R, N * N>(Sadd);
            outer_add<R, N, N>(R(1), zb, es, Lb);
            cholesky_adjoint_add<R, N>(L, Lb, Sadd);
            QM<R, N, N> Sq;
            qscatter<R, N, N>(Sadd, Sq);
            qadd<R, N, N>(Sq, Psb);
        }
    }
}

// what one backward step of the smoother reads from global memory
template <typename R, int N, int P>
struct QSmoothBwdIn {
    R m[N], mn[N], ms1[N], ms0[N], gms[N], zb[N], ab[P], es[N], eo[P];
    QM<R, N, N> Pf, Pn, An, Ps1, Ps0, gPs;
    QM<R, P, N> C;
};

template <typename R, int N, int P>
KQ_DEV void smooth_backward_quad(const SmoothBwdArgs<R>& g, int b, bool active, R* acc_LR) {
    const SmoothFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const R on = active ? R(1) : R(0);
    R msb[N];
    QM<R, N, N> Psb;
    vzero<R, N>(msb);
    qzero<R, N, N>(Psb);
    {
        QM<R, N, N> z1;
        QM<R, P, N> z2;
        R zv[N];
        qzero<R, N, N>(z1);
        qzero<R, P, N>(z2);
        vzero<R, N>(zv);
        qstore<R, N, N>(g.d_As + (long)b * N * N, z1, active);                   // A_0
        qstore<R, N, N>(g.d_As + ((long)T * B + b) * N * N, z1, active);         // A_T
        qstore<R, P, N>(g.d_Cs + ((long)(T - 1) * B + b) * P * N, z2, active);   // C_{T-1}
        vstore<R, N>(g.d_mn + ((long)(T - 1) * B + b) * N, zv, active);
        qstore<R, N, N>(g.d_Pn + ((long)(T - 1) * B + b) * N * N, z1, active);
    }
    auto load_emit = [&](long tb, QSmoothBwdIn<R, N, P>& in) {
        gload<R, N>(f.ms + tb * N, in.ms0);
        qload<R, N, N>(f.Ps + tb * N * N, in.Ps0);
        gload_or_zero<R, N>(g.g_ms, tb * N, in.gms);
        qload_or_zero<R, N, N>(g.g_Ps, tb * N * N, in.gPs);
        gload_or_zero<R, N>(g.g_zs, tb * N, in.zb);
        gload_or_zero<R, P>(g.g_ares, tb * P, in.ab);
        gload_or_zero<R, N>(f.eps_s, tb * N, in.es);
        gload_or_zero<R, P>(f.eps_o, tb * P, in.eo);
    };
    auto load_inputs = [&](int t, QSmoothBwdIn<R, N, P>& in) {
        const long tb = (long)t * B + b;
        gload<R, N>(f.m + tb * N, in.m);
        qload<R, N, N>(f.Pf + tb * N * N, in.Pf);
        gload<R, N>(f.mn + tb * N, in.mn);
        qload<R, N, N>(f.Pn + tb * N * N, in.Pn);
        qload<R, N, N>(f.As + (tb + B) * N * N, in.An);
        qload<R, P, N>(f.Cs + tb * P * N, in.C);
        gload<R, N>(f.ms + (tb + B) * N, in.ms1);
        qload<R, N, N>(f.Ps + (tb + B) * N * N, in.Ps1);
        load_emit(tb, in);
    };
    auto step = [&](int t, const QSmoothBwdIn<R, N, P>& in) {
        const long tb = (long)t * B + b;
        vadd<R, N>(in.gms, msb);
        qadd<R, N, N>(in.gPs, Psb);
        QM<R, P, N> Cb;
        qsmooth_emit_adjoint<R, N, P>(g, in.ms0, in.Ps0, in.C, in.zb, in.ab, in.es, in.eo, msb, Psb,
                                      Cb, acc_LR, on);
        qstore<R, P, N>(g.d_Cs + tb * P * N, Cb, active);
        if (t < g.burn_in) {
            vzero<R, N>(msb);
            qzero<R, N, N>(Psb);
        }

        // recompute the step
        QM<R, N, N> Pi, W, J, D;
        R d[N];
        qspd_inverse<R, N>(in.Pn, Pi);
        mm<R, N, N, N, true, false, 0>(in.An, Pi, W);
        mm<R, N, N, N, false, false, 0>(in.Pf, W, J);
        KV_UNROLL
        for (int i = 0; i < N; ++i) d[i] = in.ms1[i] - in.mn[i];
        qblend<R, N, N>(R(1), in.Ps1, R(-1), in.Pn, D);

        if (f.symmetrize) qsymmetrize<R, N>(Psb);   // H
        vstore<R, N>(g.d_m + tb * N, msb, active);
        QM<R, N, N> Pfb;
        qcopy<R, N, N>(Psb, Pfb);
        // Jbar = msb d^T + H J D^T + H^T J D
        QM<R, N, N> Jb, T1;
        qzero<R, N, N>(Jb);
        qouter_add<R, N, N>(R(1), msb, d, Jb);
        mm<R, N, N, N, false, true, 0>(J, D, T1);         // J D^T
        mm<R, N, N, N, false, false, 1>(Psb, T1, Jb);
        mm<R, N, N, N, false, false, 0>(J, D, T1);        // J D
        mm<R, N, N, N, true, false, 1>(Psb, T1, Jb);
        // dbar = J^T msb ; Dbar = J^T H J
        R db[N];
        QM<R, N, N> Db;
        mv<R, N, N, true, 0>(J, msb, db);
        mm<R, N, N, N, true, false, 0>(J, Psb, T1);
        mm<R, N, N, N, false, false, 0>(T1, J, Db);
        // J = Pf W ; W = An^T Pi
        mm<R, N, N, N, false, true, 1>(Jb, W, Pfb);       // Pfb += Jb W^T
        QM<R, N, N> Wb, Anb, Pib, Pnb;
        mm<R, N, N, N, true, false, 0>(in.Pf, Jb, Wb);    // Pf^T Jb
        mm<R, N, N, N, false, true, 0>(Pi, Wb, Anb);      // Anb = Pi Wb^T
        mm<R, N, N, N, false, false, 0>(in.An, Wb, Pib);  // Pib = An Wb
        // Pnb = -Db - Pi^T Pib Pi^T
        mm<R, N, N, N, true, false, 0>(Pi, Pib, T1);
        mm<R, N, N, N, false, true, 0>(T1, Pi, Pnb);
        qblend<R, N, N>(R(-1), Pnb, R(-1), Db, Pnb);
        R mnb[N];
        KV_UNROLL
        for (int i = 0; i < N; ++i) mnb[i] = -db[i];
        qstore<R, N, N>(g.d_P + tb * N * N, Pfb, active);
        vstore<R, N>(g.d_mn + tb * N, mnb, active);
        qstore<R, N, N>(g.d_Pn + tb * N * N, Pnb, active);
        qstore<R, N, N>(g.d_As + (tb + B) * N * N, Anb, active);
        // carry to step t+1
        vcopy<R, N>(db, msb);
        qcopy<R, N, N>(Db, Psb);
    };
    QSmoothBwdIn<R, N, P> in0, in1;
    if (T >= 2) load_inputs(0, in0);
    for (int t = 0; t < T - 1; t += 2) {
        if (t + 1 < T - 1) load_inputs(t + 1, in1);
        step(t, in0);
        if (t + 1 < T - 1) {
            if (t + 2 < T - 1) load_inputs(t + 2, in0);
            step(t + 1, in1);
        }
    }
    {
        // last step: smoothed == filtered, emission uses C_T
        const long tb = (long)(T - 1) * B + b;
        QSmoothBwdIn<R, N, P> in;
        QM<R, P, N> C, Cb;
        load_emit(tb, in);
        qload<R, P, N>(f.Cs + (tb + B) * P * N, C);
        vadd<R, N>(in.gms, msb);
        qadd<R, N, N>(in.gPs, Psb);
        qsmooth_emit_adjoint<R, N, P>(g, in.ms0, in.Ps0, C, in.zb, in.ab, in.es, in.eo, msb, Psb, Cb,
                                      acc_LR, on);
        qstore<R, P, N>(g.d_Cs + (tb + B) * P * N, Cb, active);
        vstore<R, N>(g.d_m + tb * N, msb, active);
        qstore<R, N, N>(g.d_P + tb * N * N, Psb, active);
    }
}

// Sum over the 16 sequences of the wavefront (lanes with equal q), result valid in
// the first quad.
template <typename R>
KQ_DEV R sum_over_quads(R v) {
    v += dpp<0x124>(v);   // row_ror:4 and row_ror:8 keep the lane's position in its quad
    v += dpp<0x128>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// row `row` of the partial-sum matrix (layout FilterParamGrads) from the column sums
template <typename R, int N, int P, int K>
KQ_DEV void write_filter_partial_to(const QFilterAcc<R, N, P, K>& acc, R* out) {
    using PG = FilterParamGrads<N, P, K>;
    const int lane = threadIdx.x & 63, q = lane & 3;
    KV_UNROLL
    for (int k = 0; k < K; ++k) {
        KV_UNROLL
        for (int i = 0; i < N; ++i) {
            const R v = sum_over_quads(acc.A[k][i]);
            if (lane < 4 && q < N) out[PG::OFF_A + k * N * N + i * N + q] = v;
        }
        KV_UNROLL
        for (int i = 0; i < P; ++i) {
            const R v = sum_over_quads(acc.C[k][i]);
            if (lane < 4 && q < N) out[PG::OFF_C + k * P * N + i * N + q] = v;
        }
    }
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        const R v = sum_over_quads(acc.Q[i]);
        if (lane < 4 && q < N) out[PG::OFF_Q + i * N + q] = v;
    }
    KV_UNROLL
    for (int i = 0; i < P; ++i) {
        const R v = sum_over_quads(acc.Rr[i]);
        if (lane < 4 && q < P) out[PG::OFF_R + i * P + q] = v;
    }
}
template <typename R, int N, int P, int K>
KQ_DEV void write_filter_partial(const QFilterAcc<R, N, P, K>& acc, R* partial, long row) {
    write_filter_partial_to<R, N, P, K>(acc, partial + row * FilterParamGrads<N, P, K>::LEN);
}

}  // namespace quad
}  // namespace kvae
