// Wavefront-per-sequence variant of the Kalman filter / RTS smoother / ELBO
// math and adjoints, for state dimensions whose matrices do not fit one
// lane's registers (z_dim up to 32: one covariance is 4 KiB).
//
// Execution model: ONE WAVEFRONT (= one 64-thread workgroup) OWNS ONE SEQUENCE.
// Every matrix of the sequence lives in LDS (a per-workgroup scratch struct);
// each operation below is cooperative: the 64 lanes split the output elements
// (element e is handled by lane e mod 64), read their operands from LDS and
// meet at a workgroup barrier, which for a single-wavefront workgroup costs a
// few cycles.  Values that every lane needs (mask value, pivots) are read from
// LDS by all lanes (same address = broadcast).  Sums over T and B of the
// parameter gradients are kept in registers, distributed the same way (lane
// owns elements e = lane + 64 q), and written as one row per sequence.
//
// Like kalman_core.h this is plain C++ so that tests can run the identical
// source on the host: there KW_FOR walks all elements in one thread and the
// barrier is a no-op.
//
// Algorithm sources: kvae/state_space_model.py:284-352, :390-477 and
// kvae/kalman_vae.py:191-242 of the reference; adjoints hand-derived (same
// equations as kalman_core.h).
#pragma once

#include "kalman_core.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define KW_DEVICE 1
#define KW_LANES 64
#define KW_FOR(e, total) for (int e = (int)threadIdx.x; e < (total); e += KW_LANES)
#define KW_SYNC() __syncthreads()
#define KW_LANE0 (threadIdx.x == 0)
#else
#define KW_DEVICE 0
#define KW_LANES 1
#define KW_FOR(e, total) for (int e = 0; e < (total); ++e)
#define KW_SYNC() ((void)0)
#define KW_LANE0 (true)
#endif

namespace kvae {
namespace wave {

// ---------------------------------------------------------------------------
// cooperative operations on LDS-resident row-major matrices
// ---------------------------------------------------------------------------
template <typename R>
KV_HD R wsum(R v) {
#if KW_DEVICE
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
#endif
    return v;
}

// C (M x N) {=,+=,-=} op(A) op(B), contraction length Kd (see kvae::gemm).
template <typename R, int M, int Kd, int N, bool TA, bool TB, int MODE>
KV_HD void gemm(const R* A, const R* Bm, R* C) {
    KW_FOR(e, M * N) {
        const int i = e / N, j = e % N;
        R acc = R(0);
#pragma unroll 8
        for (int k = 0; k < Kd; ++k) {
            const R a = TA ? A[k * M + i] : A[i * Kd + k];
            const R b = TB ? Bm[j * Kd + k] : Bm[k * N + j];
            acc += a * b;
        }
        if (MODE == 0) C[e] = acc;
        if (MODE == 1) C[e] += acc;
        if (MODE == 2) C[e] -= acc;
    }
    KW_SYNC();
}

template <typename R, int M, int N, bool TA, int MODE>
KV_HD void gemv(const R* A, const R* x, R* y) {
    gemm<R, M, N, 1, TA, false, MODE>(A, x, y);
}

// C (M x N) += s x y^T
template <typename R, int M, int N>
KV_HD void outer_add(R s, const R* x, const R* y, R* C) {
    KW_FOR(e, M * N) C[e] += s * x[e / N] * y[e % N];
    KW_SYNC();
}

template <typename R, int Len>
KV_HD void vzero(R* x) {
    KW_FOR(e, Len) x[e] = R(0);
    KW_SYNC();
}
template <typename R, int Len>
KV_HD void vcopy(const R* s, R* d) {
    KW_FOR(e, Len) d[e] = s[e];
    KW_SYNC();
}
template <typename R, int Len>
KV_HD void vadd(const R* s, R* d) {
    KW_FOR(e, Len) d[e] += s[e];
    KW_SYNC();
}
// d = a*x + b*y (d may alias x or y)
template <typename R, int Len>
KV_HD void axpby(R a, const R* x, R b, const R* y, R* d) {
    KW_FOR(e, Len) d[e] = a * x[e] + b * y[e];
    KW_SYNC();
}
template <typename R, int Len>
KV_HD void vscale(R a, R* x) {
    KW_FOR(e, Len) x[e] *= a;
    KW_SYNC();
}
template <typename R, int Len>
KV_HD R vdot(const R* x, const R* y) {
    R acc = R(0);
    KW_FOR(e, Len) acc += x[e] * y[e];
    return wsum<R>(acc);
}

template <typename R, int N>
KV_HD void symmetrize_inplace(R* X) {
    KW_FOR(e, N * N) {
        const int i = e / N, j = e % N;
        if (i < j) {
            const R v = (X[i * N + j] + X[j * N + i]) / R(2);
            X[i * N + j] = v;
            X[j * N + i] = v;
        }
    }
    KW_SYNC();
}

template <typename R, int Len>
KV_HD void gload(const R* p, R* x) {
    KW_FOR(e, Len) x[e] = p[e];
    KW_SYNC();
}
template <typename R, int Len>
KV_HD void gload_or_zero(const R* base, long off, R* x) {
    KW_FOR(e, Len) x[e] = base ? base[off + e] : R(0);
    KW_SYNC();
}
template <typename R, int Len>
KV_HD void gload_add(const R* base, long off, R* x) {
    if (base) {
        KW_FOR(e, Len) x[e] += base[off + e];
    }
    KW_SYNC();
}
template <typename R, int Len>
KV_HD void gstore(R* p, const R* x) {
    KW_FOR(e, Len) p[e] = x[e];
    KW_SYNC();  // x may be overwritten by other lanes next
}
template <typename R, int Len>
KV_HD void gstore_zero(R* p) {
    KW_FOR(e, Len) p[e] = R(0);
}

template <typename R, int K, int Len>
KV_HD void mix(const R* w, const R* bases, R* out) {
    KW_FOR(e, Len) {
        R acc = R(0);
#pragma unroll
        for (int k = 0; k < K; ++k) acc += w[k] * bases[k * Len + e];
        out[e] = acc;
    }
    KW_SYNC();
}

// Right-looking Cholesky, lower factor; S is destroyed (used as work space).
// Only the lower triangle of S is read.  Returns false on a pivot <= 0.
template <typename R, int N>
KV_HD bool cholesky(R* S, R* L) {
    bool ok = true;
    for (int j = 0; j < N; ++j) {
        R d = S[j * N + j];  // same address in every lane
        if (!(d > R(0))) {
            ok = false;
            d = R(1);
        }
        const R ljj = sqrt(d);
        const R inv = R(1) / ljj;
        KW_SYNC();  // everyone has read the pivot before column j is overwritten
        KW_FOR(i, N) {
            if (i < j) L[i * N + j] = R(0);
            if (i == j) L[i * N + j] = ljj;
            if (i > j) L[i * N + j] = S[i * N + j] * inv;
        }
        KW_SYNC();
        // trailing update of the lower triangle: S[i][k] -= L[i][j] L[k][j], i >= k > j
        const int rem = N - 1 - j;
        KW_FOR(e, rem * rem) {
            const int i = j + 1 + e / rem, k = j + 1 + e % rem;
            if (i >= k) S[i * N + k] -= L[i * N + j] * L[k * N + j];
        }
        KW_SYNC();
    }
    return ok;
}

// Inverse of a lower-triangular matrix; lane j builds column j.
template <typename R, int N>
KV_HD void tri_lower_inverse(const R* L, R* Li) {
    KW_FOR(j, N) {
        for (int i = 0; i < N; ++i) {
            if (i < j) {
                Li[i * N + j] = R(0);
            } else if (i == j) {
                Li[i * N + j] = R(1) / L[i * N + i];
            } else {
                R s = R(0);
                for (int k = j; k < i; ++k) s += L[i * N + k] * Li[k * N + j];
                Li[i * N + j] = -s / L[i * N + i];
            }
        }
    }
    KW_SYNC();
}

// Si = S^-1 through the Cholesky factor; S is destroyed; L and Li are outputs.
template <typename R, int N>
KV_HD bool spd_inverse(R* S, R* Si, R* L, R* Li) {
    const bool ok = cholesky<R, N>(S, L);
    tri_lower_inverse<R, N>(L, Li);
    gemm<R, N, N, N, true, false, 0>(Li, Li, Si);
    return ok;
}

// Sacc += sym( L^-T Phi(L^T tril(Lbar)) L^-1 ); X, Y are N x N work buffers.
template <typename R, int N>
KV_HD void cholesky_adjoint_add(const R* L, const R* Li, const R* Lbar, R* X, R* Y, R* Sacc) {
    KW_FOR(e, N * N) {
        const int i = e / N, j = e % N;
        R acc = R(0);
        if (i >= j) {
            for (int k = i; k < N; ++k) acc += L[k * N + i] * Lbar[k * N + j];
            if (i == j) acc *= R(0.5);
        }
        X[e] = acc;
    }
    KW_SYNC();
    gemm<R, N, N, N, true, false, 0>(Li, X, Y);   // L^-T X
    gemm<R, N, N, N, false, false, 0>(Y, Li, X);  // ... L^-1
    KW_FOR(e, N * N) {
        const int i = e / N, j = e % N;
        Sacc[e] += (X[i * N + j] + X[j * N + i]) * R(0.5);
    }
    KW_SYNC();
}

// sum_i log L[i][i]
template <typename R, int N>
KV_HD R half_logdet(const R* L) {
    R acc = R(0);
    KW_FOR(i, N) acc += log(L[i * N + i]);
    return wsum<R>(acc);
}

// Register-resident accumulator of LEN values distributed over the lanes.
template <typename R, int LEN>
struct Acc {
    static constexpr int PER = (LEN + KW_LANES - 1) / KW_LANES;
    R v[PER];
    KV_HD void zero() {
        KV_UNROLL
        for (int q = 0; q < PER; ++q) v[q] = R(0);
    }
    // this += s * x (x in LDS, LEN values)
    KV_HD void axpy(R s, const R* x) {
#if KW_DEVICE
        KV_UNROLL
        for (int q = 0; q < PER; ++q) {
            const int e = (int)threadIdx.x + KW_LANES * q;
            if (e < LEN) v[q] += s * x[e];
        }
#else
        for (int e = 0; e < LEN; ++e) v[e] += s * x[e];
#endif
    }
    KV_HD void store(R* out) const {
#if KW_DEVICE
        KV_UNROLL
        for (int q = 0; q < PER; ++q) {
            const int e = (int)threadIdx.x + KW_LANES * q;
            if (e < LEN) out[e] = v[q];
        }
#else
        for (int e = 0; e < LEN; ++e) out[e] = v[e];
#endif
    }
};

KV_HD void raise_status(int* status, int st) {
    if (st && status && KW_LANE0) {
#if KW_DEVICE
        atomicOr(status, st);
#else
        *status |= st;
#endif
    }
}

// ---------------------------------------------------------------------------
// Filter
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
struct FilterScratch {
    R Rm[P * P], Q[N * N], mp[N], Pp[N * N], m[N], Pf[N * N];
    R w[K], wn[K], a[P], C[P * N], An[N * N];
    R M[P * N], S[P * P], Si[P * P], LS[P * P], LiS[P * P], F[N * P], G[N * P], r[P];
    R T1[N * N];
};
template <typename R, int N, int P, int K>
struct FilterBwdScratch : FilterScratch<R, N, P, K> {
    R mb[N], Pb[N * N], mpb[N], Ppb[N * N], Ab[N * N], Cb[P * N], T2[N * N];
    R mob[N], Pob[N * N], Gb[N * P], rb[P], Mb[P * N], Fb[N * P], Sib[P * P], Sb[P * P],
        T3[P * P], CPt[P * N], T4[P * N], dwn[K], da[K];
};

template <typename R, int N, int P, int K>
KV_HD bool update_terms(FilterScratch<R, N, P, K>& s) {
    gemm<R, P, N, N, false, false, 0>(s.C, s.Pp, s.M);
    vcopy<R, P * P>(s.Rm, s.S);
    gemm<R, P, N, P, false, true, 1>(s.M, s.C, s.S);
    const bool ok = spd_inverse<R, P>(s.S, s.Si, s.LS, s.LiS);
    gemm<R, N, N, P, false, true, 0>(s.Pp, s.C, s.F);
    gemm<R, N, P, P, false, false, 0>(s.F, s.Si, s.G);
    vcopy<R, P>(s.a, s.r);
    gemv<R, P, N, false, 2>(s.C, s.mp, s.r);
    return ok;
}

template <typename R, int N, int P, int K>
KV_HD void filter_forward_sequence(const FilterFwdArgs<R>& g, int b, FilterScratch<R, N, P, K>& s) {
    const long B = g.B;
    gload<R, P * P>(g.Rm, s.Rm);
    gload<R, N * N>(g.Q, s.Q);
    if (g.carry_per_sequence) {
        gload<R, N>(g.m_in + (long)b * N, s.mp);
        gload<R, N * N>(g.P_in + (long)b * N * N, s.Pp);
    } else {
        gload<R, N>(g.m_in, s.mp);
        gload<R, N * N>(g.P_in, s.Pp);
    }
    gload<R, K>(g.alpha + (long)b * K, s.w);
    if (g.As) {
        mix<R, K, N * N>(s.w, g.A_K, s.An);
        gstore<R, N * N>(g.As + (long)b * N * N, s.An);
    }
    int st = 0;
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        gload<R, K>(g.alpha + (tb + B) * K, s.wn);
        gload<R, P>(g.a + tb * P, s.a);
        mix<R, K, P * N>(s.w, g.C_K, s.C);
        mix<R, K, N * N>(s.wn, g.A_K, s.An);
        if (g.Cs) gstore<R, P * N>(g.Cs + tb * P * N, s.C);
        if (g.As) gstore<R, N * N>(g.As + (tb + B) * N * N, s.An);
        if (!update_terms<R, N, P, K>(s)) st |= ST_NONPD_INNOVATION;

        vcopy<R, N>(s.mp, s.m);
        gemv<R, N, P, false, 1>(s.G, s.r, s.m);
        vcopy<R, N * N>(s.Pp, s.Pf);
        gemm<R, N, P, N, false, false, 2>(s.G, s.M, s.Pf);
        if (g.mask) {
            const R mu = g.mask[tb];
            axpby<R, N>(mu, s.m, R(1) - mu, s.mp, s.m);
            axpby<R, N * N>(mu, s.Pf, R(1) - mu, s.Pp, s.Pf);
        }
        if (g.symmetrize) symmetrize_inplace<R, N>(s.Pf);
        gemv<R, N, N, false, 0>(s.An, s.m, s.mp);
        gemm<R, N, N, N, false, false, 0>(s.An, s.Pf, s.T1);
        vcopy<R, N * N>(s.Q, s.Pp);
        gemm<R, N, N, N, false, true, 1>(s.T1, s.An, s.Pp);
        if (g.symmetrize) symmetrize_inplace<R, N>(s.Pp);
        gstore<R, N>(g.m + tb * N, s.m);
        gstore<R, N * N>(g.Pf + tb * N * N, s.Pf);
        gstore<R, N>(g.mn + tb * N, s.mp);
        gstore<R, N * N>(g.Pn + tb * N * N, s.Pp);
        vcopy<R, K>(s.wn, s.w);
    }
    if (g.Cs) {
        mix<R, K, P * N>(s.w, g.C_K, s.C);
        gstore<R, P * N>(g.Cs + ((long)g.T * B + b) * P * N, s.C);
    }
    raise_status(g.status, st);
}

template <typename R, int N, int P, int K>
struct FilterGradAcc {
    Acc<R, K * N * N> dA;
    Acc<R, K * P * N> dC;
    Acc<R, N * N> dQ;
    Acc<R, P * P> dR;
    KV_HD void zero() {
        dA.zero();
        dC.zero();
        dQ.zero();
        dR.zero();
    }
    KV_HD void store(R* row) const {
        using PG = FilterParamGrads<N, P, K>;
        dA.store(row + PG::OFF_A);
        dC.store(row + PG::OFF_C);
        dQ.store(row + PG::OFF_Q);
        dR.store(row + PG::OFF_R);
    }
};

// acc.dX[k*Len + e] += w[k] * x[e]
template <typename R, int K, int Len>
KV_HD void acc_mix(Acc<R, K * Len>& acc, const R* w, const R* x) {
#if KW_DEVICE
    constexpr int PER = Acc<R, K * Len>::PER;
    KV_UNROLL
    for (int q = 0; q < PER; ++q) {
        const int idx = (int)threadIdx.x + KW_LANES * q;
        if (idx < K * Len) acc.v[q] += w[idx / Len] * x[idx % Len];
    }
#else
    for (int idx = 0; idx < K * Len; ++idx) acc.v[idx] += w[idx / Len] * x[idx % Len];
#endif
}

// out[k] = <x, bases[k]> for k < K (bases in global memory)
template <typename R, int K, int Len>
KV_HD void basis_dots(const R* x, const R* bases, R* out) {
    for (int k = 0; k < K; ++k) {
        R acc = R(0);
        KW_FOR(e, Len) acc += x[e] * bases[k * Len + e];
        acc = wsum<R>(acc);
        if (KW_LANE0) out[k] = acc;
    }
    KW_SYNC();
}

template <typename R, int N, int P, int K>
KV_HD void filter_backward_sequence(const FilterBwdArgs<R>& g, int b,
                                    FilterBwdScratch<R, N, P, K>& s,
                                    FilterGradAcc<R, N, P, K>& acc) {
    const FilterFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    gload<R, P * P>(f.Rm, s.Rm);
    vzero<R, N>(s.mpb);
    vzero<R, N * N>(s.Ppb);
    {
        gload_or_zero<R, P * N>(g.g_Cs, ((long)T * B + b) * P * N, s.Cb);
        gload<R, K>(f.alpha + ((long)T * B + b) * K, s.w);
        basis_dots<R, K, P * N>(s.Cb, f.C_K, s.dwn);
        acc_mix<R, K, P * N>(acc.dC, s.w, s.Cb);
    }
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        gload<R, K>(f.alpha + tb * K, s.w);
        gload<R, K>(f.alpha + (tb + B) * K, s.wn);
        gload<R, P>(f.a + tb * P, s.a);
        mix<R, K, P * N>(s.w, f.C_K, s.C);
        mix<R, K, N * N>(s.wn, f.A_K, s.An);
        if (t > 0) {
            gload<R, N>(f.mn + (tb - B) * N, s.mp);
            gload<R, N * N>(f.Pn + (tb - B) * N * N, s.Pp);
        } else if (f.carry_per_sequence) {
            gload<R, N>(f.m_in + (long)b * N, s.mp);
            gload<R, N * N>(f.P_in + (long)b * N * N, s.Pp);
        } else {
            gload<R, N>(f.m_in, s.mp);
            gload<R, N * N>(f.P_in, s.Pp);
        }
        gload<R, N>(f.m + tb * N, s.m);
        gload<R, N * N>(f.Pf + tb * N * N, s.Pf);

        gload_or_zero<R, N>(g.g_m, tb * N, s.mb);
        gload_or_zero<R, N * N>(g.g_P, tb * N * N, s.Pb);
        gload_add<R, N>(g.g_mn, tb * N, s.mpb);
        gload_add<R, N * N>(g.g_Pn, tb * N * N, s.Ppb);
        if (t < g.burn_in) {
            vzero<R, N>(s.mb);
            vzero<R, N * N>(s.Pb);
            vzero<R, N>(s.mpb);
            vzero<R, N * N>(s.Ppb);
        }
        // time update adjoint
        gload_or_zero<R, N * N>(g.g_As, (tb + B) * N * N, s.Ab);
        if (f.symmetrize) symmetrize_inplace<R, N>(s.Ppb);
        outer_add<R, N, N>(R(1), s.mpb, s.m, s.Ab);
        gemm<R, N, N, N, false, true, 0>(s.An, s.Pf, s.T1);
        gemm<R, N, N, N, false, false, 1>(s.Ppb, s.T1, s.Ab);
        gemm<R, N, N, N, false, false, 0>(s.An, s.Pf, s.T1);
        gemm<R, N, N, N, true, false, 1>(s.Ppb, s.T1, s.Ab);
        gemv<R, N, N, true, 1>(s.An, s.mpb, s.mb);
        gemm<R, N, N, N, true, false, 0>(s.An, s.Ppb, s.T2);
        gemm<R, N, N, N, false, false, 1>(s.T2, s.An, s.Pb);
        acc.dQ.axpy(R(1), s.Ppb);

        // measurement update adjoint
        update_terms<R, N, P, K>(s);
        const R mu = f.mask ? f.mask[tb] : R(1);
        const R nu = R(1) - mu;
        if (f.symmetrize) symmetrize_inplace<R, N>(s.Pb);
        gload_or_zero<R, P * N>(g.g_Cs, tb * P * N, s.Cb);
        KW_FOR(i, N) {
            s.mob[i] = mu * s.mb[i];
            s.mpb[i] = nu * s.mb[i] + mu * s.mb[i];
        }
        KW_FOR(i, N * N) {
            s.Pob[i] = mu * s.Pb[i];
            s.Ppb[i] = nu * s.Pb[i] + mu * s.Pb[i];
        }
        KW_SYNC();
        vzero<R, N * P>(s.Gb);
        outer_add<R, N, P>(R(1), s.mob, s.r, s.Gb);
        gemv<R, P, N, true, 0>(s.G, s.mob, s.rb);
        gstore<R, P>(g.d_a + tb * P, s.rb);
        outer_add<R, P, N>(R(-1), s.rb, s.mp, s.Cb);
        gemv<R, N, P, true, 2>(s.C, s.rb, s.mpb);
        gemm<R, N, N, P, false, true, 2>(s.Pob, s.M, s.Gb);
        gemm<R, P, N, N, true, false, 0>(s.G, s.Pob, s.Mb);
        gemm<R, P, N, N, false, true, 2>(s.Mb, s.Pp, s.Cb);
        gemm<R, N, P, N, true, false, 2>(s.C, s.Mb, s.Ppb);
        gemm<R, N, P, P, false, true, 0>(s.Gb, s.Si, s.Fb);
        gemm<R, P, N, P, true, false, 0>(s.F, s.Gb, s.Sib);
        gemm<R, P, P, P, true, false, 0>(s.Si, s.Sib, s.T3);
        gemm<R, P, P, P, false, true, 0>(s.T3, s.Si, s.Sb);
        vscale<R, P * P>(R(-1), s.Sb);
        gemm<R, N, P, N, false, false, 1>(s.Fb, s.C, s.Ppb);
        gemm<R, P, N, N, true, false, 1>(s.Fb, s.Pp, s.Cb);
        gemm<R, P, N, N, false, true, 0>(s.C, s.Pp, s.CPt);
        gemm<R, P, P, N, false, false, 1>(s.Sb, s.CPt, s.Cb);
        gemm<R, P, P, N, true, false, 1>(s.Sb, s.M, s.Cb);
        gemm<R, P, P, N, false, false, 0>(s.Sb, s.C, s.T4);
        gemm<R, N, P, N, true, false, 1>(s.C, s.T4, s.Ppb);
        acc.dR.axpy(R(1), s.Sb);

        // mixing adjoint
        basis_dots<R, K, N * N>(s.Ab, f.A_K, s.da);
        KW_FOR(k, K) s.da[k] += s.dwn[k];
        KW_SYNC();
        gstore<R, K>(g.d_alpha + (tb + B) * K, s.da);
        acc_mix<R, K, N * N>(acc.dA, s.wn, s.Ab);
        basis_dots<R, K, P * N>(s.Cb, f.C_K, s.dwn);
        acc_mix<R, K, P * N>(acc.dC, s.w, s.Cb);
    }
    {
        gload_or_zero<R, N * N>(g.g_As, (long)b * N * N, s.Ab);
        gload<R, K>(f.alpha + (long)b * K, s.w);
        basis_dots<R, K, N * N>(s.Ab, f.A_K, s.da);
        KW_FOR(k, K) s.da[k] += s.dwn[k];
        KW_SYNC();
        gstore<R, K>(g.d_alpha + (long)b * K, s.da);
        acc_mix<R, K, N * N>(acc.dA, s.w, s.Ab);
    }
    gstore<R, N>(g.d_m_in + (long)b * N, s.mpb);
    gstore<R, N * N>(g.d_P_in + (long)b * N * N, s.Ppb);
}

// ---------------------------------------------------------------------------
// Smoother
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
struct SmoothScratch {
    R LR[P * P], ms[N], Ps[N * N], C[P * N], m[N], Pf[N * N], mn[N], Pn[N * N], An[N * N];
    R Pi[N * N], L[N * N], Li[N * N], W[N * N], J[N * N], d[N], D[N * N], T1[N * N];
    R z[N], ar[P], e[N], eo[P];
};
template <typename R, int N, int P>
struct SmoothBwdScratch : SmoothScratch<R, N, P> {
    R ms1[N], Ps1[N * N], msb[N], Psb[N * N], Cb[P * N], zb[N], ab[P], Lb[N * N], X[N * N],
        Y[N * N], Pfb[N * N], Jb[N * N], db[N], Db[N * N], Wb[N * N], Anb[N * N], Pib[N * N],
        Pnb[N * N];
};

template <typename R, int N, int P>
KV_HD bool smooth_emit(const SmoothFwdArgs<R>& g, long tb, SmoothScratch<R, N, P>& s) {
    bool ok = true;
    vcopy<R, N>(s.ms, s.z);
    if (g.eps_s) {
        vcopy<R, N * N>(s.Ps, s.T1);
        ok = cholesky<R, N>(s.T1, s.L);
        gload<R, N>(g.eps_s + tb * N, s.e);
        gemv<R, N, N, false, 1>(s.L, s.e, s.z);
    }
    gemv<R, P, N, false, 0>(s.C, s.z, s.ar);
    if (g.eps_o) {
        gload<R, P>(g.eps_o + tb * P, s.eo);
        gemv<R, P, P, false, 1>(s.LR, s.eo, s.ar);
    }
    gstore<R, N>(g.zs + tb * N, s.z);
    gstore<R, P>(g.ares + tb * P, s.ar);
    return ok;
}

// J = Pf An^T Pn^-1, d = ms_next - mn, D = Ps_next - Pn; (ms, Ps) hold step t+1 on entry.
template <typename R, int N, int P>
KV_HD bool smooth_gain(SmoothScratch<R, N, P>& s, const R* ms_next, const R* Ps_next) {
    vcopy<R, N * N>(s.Pn, s.T1);
    const bool ok = spd_inverse<R, N>(s.T1, s.Pi, s.L, s.Li);
    gemm<R, N, N, N, true, false, 0>(s.An, s.Pi, s.W);
    gemm<R, N, N, N, false, false, 0>(s.Pf, s.W, s.J);
    axpby<R, N>(R(1), ms_next, R(-1), s.mn, s.d);
    axpby<R, N * N>(R(1), Ps_next, R(-1), s.Pn, s.D);
    return ok;
}

template <typename R, int N, int P>
KV_HD void smooth_forward_sequence(const SmoothFwdArgs<R>& g, int b, SmoothScratch<R, N, P>& s) {
    const long B = g.B;
    const int T = g.T;
    if (g.eps_o) {
        gload<R, P * P>(g.LR, s.LR);
    }
    int st = 0;
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(g.m + tb * N, s.ms);
        gload<R, N * N>(g.Pf + tb * N * N, s.Ps);
        gload<R, P * N>(g.Cs + (tb + B) * P * N, s.C);
        gstore<R, N>(g.ms + tb * N, s.ms);
        gstore<R, N * N>(g.Ps + tb * N * N, s.Ps);
        if (!smooth_emit<R, N, P>(g, tb, s)) st |= ST_NONPD_SMOOTHED;
    }
    for (int t = T - 2; t >= 0; --t) {
        const long tb = (long)t * B + b;
        gload<R, N>(g.m + tb * N, s.m);
        gload<R, N * N>(g.Pf + tb * N * N, s.Pf);
        gload<R, N>(g.mn + tb * N, s.mn);
        gload<R, N * N>(g.Pn + tb * N * N, s.Pn);
        gload<R, N * N>(g.As + (tb + B) * N * N, s.An);
        gload<R, P * N>(g.Cs + tb * P * N, s.C);
        if (!smooth_gain<R, N, P>(s, s.ms, s.Ps)) st |= ST_NONPD_PREDICTED;
        vcopy<R, N>(s.m, s.ms);
        gemv<R, N, N, false, 1>(s.J, s.d, s.ms);
        gemm<R, N, N, N, false, false, 0>(s.J, s.D, s.T1);
        vcopy<R, N * N>(s.Pf, s.Ps);
        gemm<R, N, N, N, false, true, 1>(s.T1, s.J, s.Ps);
        if (g.symmetrize) symmetrize_inplace<R, N>(s.Ps);
        gstore<R, N>(g.ms + tb * N, s.ms);
        gstore<R, N * N>(g.Ps + tb * N * N, s.Ps);
        if (!smooth_emit<R, N, P>(g, tb, s)) st |= ST_NONPD_SMOOTHED;
    }
    raise_status(g.status, st);
}

// adjoint of the emission at pair tb for the smoothed (ms, Ps) in s.ms/s.Ps and C in s.C:
// adds into s.msb / s.Psb, writes s.Cb.
template <typename R, int N, int P>
KV_HD void smooth_emit_adjoint(const SmoothBwdArgs<R>& g, long tb, SmoothBwdScratch<R, N, P>& s,
                               Acc<R, P * P>& acc_LR) {
    const SmoothFwdArgs<R>& f = g.f;
    gload_or_zero<R, N>(g.g_zs, tb * N, s.zb);
    gload_or_zero<R, P>(g.g_ares, tb * P, s.ab);
    vzero<R, P * N>(s.Cb);
    if (!g.g_ares && !g.g_zs) return;
    if (f.eps_s) {
        vcopy<R, N * N>(s.Ps, s.T1);
        cholesky<R, N>(s.T1, s.L);
        gload<R, N>(f.eps_s + tb * N, s.e);
    }
    if (g.g_ares) {
        vcopy<R, N>(s.ms, s.z);
        if (f.eps_s) gemv<R, N, N, false, 1>(s.L, s.e, s.z);
        outer_add<R, P, N>(R(1), s.ab, s.z, s.Cb);
        gemv<R, N, P, true, 1>(s.C, s.ab, s.zb);
        if (f.eps_o) {
            gload<R, P>(f.eps_o + tb * P, s.eo);
            vzero<R, P * P>(s.X);
            outer_add<R, P, P>(R(1), s.ab, s.eo, s.X);
            acc_LR.axpy(R(1), s.X);
        }
    }
    vadd<R, N>(s.zb, s.msb);
    if (f.eps_s) {
        tri_lower_inverse<R, N>(s.L, s.Li);
        vzero<R, N * N>(s.Lb);
        outer_add<R, N, N>(R(1), s.zb, s.e, s.Lb);
        cholesky_adjoint_add<R, N>(s.L, s.Li, s.Lb, s.X, s.Y, s.Psb);
    }
}

template <typename R, int N, int P>
KV_HD void smooth_backward_sequence(const SmoothBwdArgs<R>& g, int b,
                                    SmoothBwdScratch<R, N, P>& s,
                                    Acc<R, P * P>& acc_LR) {
    const SmoothFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    vzero<R, N>(s.msb);
    vzero<R, N * N>(s.Psb);
    gstore_zero<R, N * N>(g.d_As + (long)b * N * N);
    gstore_zero<R, N * N>(g.d_As + ((long)T * B + b) * N * N);
    gstore_zero<R, P * N>(g.d_Cs + ((long)(T - 1) * B + b) * P * N);
    gstore_zero<R, N>(g.d_mn + ((long)(T - 1) * B + b) * N);
    gstore_zero<R, N * N>(g.d_Pn + ((long)(T - 1) * B + b) * N * N);
    for (int t = 0; t < T - 1; ++t) {
        const long tb = (long)t * B + b;
        gload<R, N>(f.m + tb * N, s.m);
        gload<R, N * N>(f.Pf + tb * N * N, s.Pf);
        gload<R, N>(f.mn + tb * N, s.mn);
        gload<R, N * N>(f.Pn + tb * N * N, s.Pn);
        gload<R, N * N>(f.As + (tb + B) * N * N, s.An);
        gload<R, P * N>(f.Cs + tb * P * N, s.C);
        gload<R, N>(f.ms + (tb + B) * N, s.ms1);
        gload<R, N * N>(f.Ps + (tb + B) * N * N, s.Ps1);
        gload<R, N>(f.ms + tb * N, s.ms);
        gload<R, N * N>(f.Ps + tb * N * N, s.Ps);
        gload_add<R, N>(g.g_ms, tb * N, s.msb);
        gload_add<R, N * N>(g.g_Ps, tb * N * N, s.Psb);
        smooth_emit_adjoint<R, N, P>(g, tb, s, acc_LR);
        gstore<R, P * N>(g.d_Cs + tb * P * N, s.Cb);
        if (t < g.burn_in) {
            vzero<R, N>(s.msb);
            vzero<R, N * N>(s.Psb);
        }
        smooth_gain<R, N, P>(s, s.ms1, s.Ps1);
        if (f.symmetrize) symmetrize_inplace<R, N>(s.Psb);
        gstore<R, N>(g.d_m + tb * N, s.msb);
        vcopy<R, N * N>(s.Psb, s.Pfb);
        vzero<R, N * N>(s.Jb);
        outer_add<R, N, N>(R(1), s.msb, s.d, s.Jb);
        gemm<R, N, N, N, false, true, 0>(s.J, s.D, s.T1);
        gemm<R, N, N, N, false, false, 1>(s.Psb, s.T1, s.Jb);
        gemm<R, N, N, N, false, false, 0>(s.J, s.D, s.T1);
        gemm<R, N, N, N, true, false, 1>(s.Psb, s.T1, s.Jb);
        gemv<R, N, N, true, 0>(s.J, s.msb, s.db);
        gemm<R, N, N, N, true, false, 0>(s.J, s.Psb, s.T1);
        gemm<R, N, N, N, false, false, 0>(s.T1, s.J, s.Db);
        gemm<R, N, N, N, false, true, 1>(s.Jb, s.W, s.Pfb);
        gemm<R, N, N, N, true, false, 0>(s.Pf, s.Jb, s.Wb);
        gemm<R, N, N, N, false, true, 0>(s.Pi, s.Wb, s.Anb);
        gemm<R, N, N, N, false, false, 0>(s.An, s.Wb, s.Pib);
        gemm<R, N, N, N, true, false, 0>(s.Pi, s.Pib, s.T1);
        gemm<R, N, N, N, false, true, 0>(s.T1, s.Pi, s.Pnb);
        axpby<R, N * N>(R(-1), s.Pnb, R(-1), s.Db, s.Pnb);
        gstore<R, N * N>(g.d_P + tb * N * N, s.Pfb);
        KW_FOR(i, N) g.d_mn[tb * N + i] = -s.db[i];
        gstore<R, N * N>(g.d_Pn + tb * N * N, s.Pnb);
        gstore<R, N * N>(g.d_As + (tb + B) * N * N, s.Anb);
        vcopy<R, N>(s.db, s.msb);
        vcopy<R, N * N>(s.Db, s.Psb);
    }
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(f.ms + tb * N, s.ms);
        gload<R, N * N>(f.Ps + tb * N * N, s.Ps);
        gload<R, P * N>(f.Cs + (tb + B) * P * N, s.C);
        gload_add<R, N>(g.g_ms, tb * N, s.msb);
        gload_add<R, N * N>(g.g_Ps, tb * N * N, s.Psb);
        smooth_emit_adjoint<R, N, P>(g, tb, s, acc_LR);
        gstore<R, P * N>(g.d_Cs + (tb + B) * P * N, s.Cb);
        gstore<R, N>(g.d_m + tb * N, s.msb);
        gstore<R, N * N>(g.d_P + tb * N * N, s.Psb);
    }
}

// ---------------------------------------------------------------------------
// ELBO terms, one wavefront per sequence, sequential in t.
// Forward runs t = 0..T-1 keeping z_{t-1}; backward runs t = T-1..0 keeping
// the next pair's transition gradient direction A_{t+1}^T u_{t+1}.
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
struct ElboScratch {
    R Q[N * N], LQ[N * N], LiQ[N * N], Rm[P * P], LRm[P * P], LiR[P * P];
    R P0[N * N], L0[N * N], Li0[N * N], m0[N];
    R Ps[N * N], L[N * N], Li[N * N], z[N], zp[N], e[N], C[P * N], A[N * N], r[N], ro[P], y[N];
    R T1[N * N];
};
template <typename R, int N, int P>
struct ElboBwdScratch : ElboScratch<R, N, P> {
    R zn_back[N], zb[N], Ab[N * N], Cb[P * N], Lb[N * N], X[N * N], Y[N * N], Psb[N * N],
        SiQ[N * N], SiR[P * P], Si0[N * N], G1[N * N], G2[P * P];
};

// s.z = ms + chol(Ps) eps at pair tb; leaves L in s.L, eps in s.e
template <typename R, int N, int P>
KV_HD bool elbo_sample(const ElboFwdArgs<R>& g, long tb, ElboScratch<R, N, P>& s) {
    gload<R, N * N>(g.Ps + tb * N * N, s.T1);
    gload<R, N>(g.ms + tb * N, s.z);
    gload<R, N>(g.eps + tb * N, s.e);
    const bool ok = cholesky<R, N>(s.T1, s.L);
    gemv<R, N, N, false, 1>(s.L, s.e, s.z);
    return ok;
}

// r := Sigma^-1 r via Li (y = Li r, r = Li^T y); returns |y|^2
template <typename R, int D>
KV_HD R whiten(const R* Li, R* r, R* y) {
    gemv<R, D, D, false, 0>(Li, r, y);
    const R q = vdot<R, D>(y, y);
    gemv<R, D, D, true, 0>(Li, y, r);
    return q;
}

template <typename R, int N, int P>
KV_HD int elbo_prepare(const ElboFwdArgs<R>& g, ElboScratch<R, N, P>& s) {
    int st = 0;
    gload<R, N * N>(g.Q, s.Q);
    vcopy<R, N * N>(s.Q, s.T1);
    if (!cholesky<R, N>(s.T1, s.LQ)) st |= ST_NONPD_NOISE;
    tri_lower_inverse<R, N>(s.LQ, s.LiQ);
    gload<R, P * P>(g.Rm, s.Rm);
    vcopy<R, P * P>(s.Rm, s.T1);
    if (!cholesky<R, P>(s.T1, s.LRm)) st |= ST_NONPD_NOISE;
    tri_lower_inverse<R, P>(s.LRm, s.LiR);
    gload<R, N * N>(g.P0, s.P0);
    vcopy<R, N * N>(s.P0, s.T1);
    if (!cholesky<R, N>(s.T1, s.L0)) st |= ST_NONPD_NOISE;
    tri_lower_inverse<R, N>(s.L0, s.Li0);
    gload<R, N>(g.m0, s.m0);
    return st;
}

template <typename R, int N, int P>
KV_HD void elbo_forward_sequence(const ElboFwdArgs<R>& g, int b, ElboScratch<R, N, P>& s) {
    const long B = g.B;
    int st = elbo_prepare<R, N, P>(g, s);
    const R hlQ = half_logdet<R, N>(s.LQ), hlR = half_logdet<R, P>(s.LRm),
            hl0 = half_logdet<R, N>(s.L0);
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        if (!elbo_sample<R, N, P>(g, tb, s)) st |= ST_NONPD_SMOOTHED;
        if (g.z) gstore<R, N>(g.z + tb * N, s.z);
        // observation
        gload<R, P * N>(g.Cs + tb * P * N, s.C);
        gload<R, P>(g.a + tb * P, s.ro);
        gemv<R, P, N, false, 2>(s.C, s.z, s.ro);
        const R qo = whiten<R, P>(s.LiR, s.ro, s.y);
        // transition
        R qt, hlt;
        if (t == 0) {
            axpby<R, N>(R(1), s.z, R(-1), s.m0, s.r);
            qt = whiten<R, N>(s.Li0, s.r, s.y);
            hlt = hl0;
        } else {
            gload<R, N * N>(g.As + tb * N * N, s.A);
            vcopy<R, N>(s.z, s.r);
            gemv<R, N, N, false, 2>(s.A, s.zp, s.r);
            qt = whiten<R, N>(s.LiQ, s.r, s.y);
            hlt = hlQ;
        }
        const R qp = vdot<R, N>(s.e, s.e);
        const R hlp = half_logdet<R, N>(s.L);
        if (KW_LANE0) {
            g.obs[tb] = R(-0.5) * (R(P) * log_2pi<R>() + qo) - hlR;
            g.trans[tb] = R(-0.5) * (R(N) * log_2pi<R>() + qt) - hlt;
            g.post[tb] = R(-0.5) * (R(N) * log_2pi<R>() + qp) - hlp;
        }
        vcopy<R, N>(s.z, s.zp);
    }
    raise_status(g.status, st);
}

template <typename R, int N, int P>
struct ElboGradAcc {
    Acc<R, N * N> dQ;
    Acc<R, P * P> dR;
    Acc<R, N> dm0;
    Acc<R, N * N> dP0;
    KV_HD void zero() {
        dQ.zero();
        dR.zero();
        dm0.zero();
        dP0.zero();
    }
    KV_HD void store(R* row) const {
        using PG = ElboParamGrads<N, P>;
        dQ.store(row + PG::OFF_Q);
        dR.store(row + PG::OFF_R);
        dm0.store(row + PG::OFF_M0);
        dP0.store(row + PG::OFF_P0);
    }
};

// G (D x D work buffer) = 1/2 (u u^T - Si); acc += w * G
template <typename R, int D>
KV_HD void cov_grad(R w, const R* u, const R* Si, R* G, Acc<R, D * D>& acc) {
    KW_FOR(e, D * D) G[e] = R(0.5) * (u[e / D] * u[e % D] - Si[e]);
    KW_SYNC();
    acc.axpy(w, G);
}

template <typename R, int N, int P>
KV_HD void elbo_backward_sequence(const ElboBwdArgs<R>& g, int b, ElboBwdScratch<R, N, P>& s,
                                  ElboGradAcc<R, N, P>& acc) {
    const ElboFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    elbo_prepare<R, N, P>(f, s);
    gemm<R, N, N, N, true, false, 0>(s.LiQ, s.LiQ, s.SiQ);
    gemm<R, P, P, P, true, false, 0>(s.LiR, s.LiR, s.SiR);
    gemm<R, N, N, N, true, false, 0>(s.Li0, s.Li0, s.Si0);
    gstore_zero<R, N * N>(g.d_As + ((long)T * B + b) * N * N);
    gstore_zero<R, P * N>(g.d_Cs + ((long)T * B + b) * P * N);
    vzero<R, N>(s.zn_back);  // w_{t+1} A_{t+1}^T u_{t+1}, zero for t = T-1
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        const R wo = g.g_obs ? g.g_obs[tb] : R(0);
        const R wt = g.g_trans ? g.g_trans[tb] : R(0);
        const R wp = g.g_post ? g.g_post[tb] : R(0);
        elbo_sample<R, N, P>(f, tb, s);  // s.z, s.L, s.e
        vcopy<R, N>(s.zn_back, s.zb);
        // observation term
        gload<R, P * N>(f.Cs + tb * P * N, s.C);
        gload<R, P>(f.a + tb * P, s.ro);
        gemv<R, P, N, false, 2>(s.C, s.z, s.ro);
        whiten<R, P>(s.LiR, s.ro, s.y);  // s.ro = R^-1 r
        KW_FOR(i, P) g.d_a[tb * P + i] = -wo * s.ro[i];
        vzero<R, P * N>(s.Cb);
        outer_add<R, P, N>(wo, s.ro, s.z, s.Cb);
        gstore<R, P * N>(g.d_Cs + tb * P * N, s.Cb);
        KW_FOR(j, N) {
            R acc_j = R(0);
            for (int i = 0; i < P; ++i) acc_j += s.C[i * N + j] * s.ro[i];
            s.zb[j] += wo * acc_j;
        }
        KW_SYNC();
        cov_grad<R, P>(wo, s.ro, s.SiR, s.G2, acc.dR);
        // transition term of this pair
        vzero<R, N * N>(s.Ab);
        if (t == 0) {
            axpby<R, N>(R(1), s.z, R(-1), s.m0, s.r);
            whiten<R, N>(s.Li0, s.r, s.y);
            acc.dm0.axpy(wt, s.r);
            cov_grad<R, N>(wt, s.r, s.Si0, s.G1, acc.dP0);
            vzero<R, N>(s.zn_back);
        } else {
            // previous sample z_{t-1}: recompute into s.zp without disturbing s.L / s.e
            {
                const long tp = tb - B;
                gload<R, N * N>(f.Ps + tp * N * N, s.T1);
                gload<R, N>(f.ms + tp * N, s.zp);
                gload<R, N>(f.eps + tp * N, s.y);
                cholesky<R, N>(s.T1, s.X);
                gemv<R, N, N, false, 1>(s.X, s.y, s.zp);
            }
            gload<R, N * N>(f.As + tb * N * N, s.A);
            vcopy<R, N>(s.z, s.r);
            gemv<R, N, N, false, 2>(s.A, s.zp, s.r);
            whiten<R, N>(s.LiQ, s.r, s.y);  // s.r = Q^-1 r
            outer_add<R, N, N>(wt, s.r, s.zp, s.Ab);
            cov_grad<R, N>(wt, s.r, s.SiQ, s.G1, acc.dQ);
            // contribution to z_{t-1}: w_t A_t^T u_t, consumed by the next iteration
            KW_FOR(j, N) {
                R acc_j = R(0);
                for (int i = 0; i < N; ++i) acc_j += s.A[i * N + j] * s.r[i];
                s.zn_back[j] = wt * acc_j;
            }
            KW_SYNC();
        }
        KW_FOR(i, N) s.zb[i] -= wt * s.r[i];
        KW_SYNC();
        gstore<R, N * N>(g.d_As + tb * N * N, s.Ab);
        gstore<R, N>(g.d_ms + tb * N, s.zb);
        // z = ms + L eps; posterior term: -wp sum log diag L
        vzero<R, N * N>(s.Lb);
        outer_add<R, N, N>(R(1), s.zb, s.e, s.Lb);
        KW_FOR(i, N) s.Lb[i * N + i] -= wp / s.L[i * N + i];
        KW_SYNC();
        tri_lower_inverse<R, N>(s.L, s.Li);
        vzero<R, N * N>(s.Psb);
        cholesky_adjoint_add<R, N>(s.L, s.Li, s.Lb, s.X, s.Y, s.Psb);
        gstore<R, N * N>(g.d_Ps + tb * N * N, s.Psb);
    }
}

}  // namespace wave
}  // namespace kvae
