// Per-sequence math of the Kalman filter, RTS smoother and Kalman ELBO terms
// and of their hand-derived adjoints, for small state dimensions.
//
// Execution model: ONE LANE OWNS ONE SEQUENCE.  Every n x n / p x n / p x p
// matrix of that sequence lives in the lane's registers; all loops over
// matrix indices have compile-time bounds and are fully unrolled, so no array
// below is ever indexed with a run-time value (that would push it to scratch
// memory).  The loop over time steps is the only real loop.  Because lane b
// reads element [t][b][...] of every (T,B,...) tensor, a wavefront touches one
// contiguous 64 * row_bytes span per tensor and step.
//
// The functions are plain C++ templates so that the identical source can be
// compiled for the device (kernels.hip) and, for tests only, for the host.
//
// Algorithm sources (reference checkout): kvae/state_space_model.py:284-352
// (filter step), :424-472 (smoother step), kvae/kalman_vae.py:191-242 (ELBO
// terms).  The reference differentiates these with the autograd tape; the
// adjoints here are derived by hand.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define KV_HD __host__ __device__ __forceinline__
#else
#define KV_HD inline
#endif

#define KV_UNROLL _Pragma("unroll")

namespace kvae {

enum : int {
    ST_NONPD_INNOVATION = 1,
    ST_NONPD_PREDICTED = 2,
    ST_NONPD_SMOOTHED = 4,
    ST_NONPD_NOISE = 8,
};

// ---------------------------------------------------------------------------
// Reciprocal, reciprocal square root and natural logarithm of a pivot.
// fp32 device code uses the hardware approximations (v_rcp_f32, v_rsq_f32,
// v_log_f32: 1 ulp on normal numbers, one instruction each) instead of the
// correctly rounded division / sqrt / log sequences, which are 10-15 dependent
// instructions around a quarter-rate operation and sat on the serial chain of
// every time step.  Pivots are O(1) positive numbers (a non-positive pivot is
// replaced by 1 before it gets here), so the denormal and range fix-ups those
// sequences add are not needed.  fp64 and the host build keep IEEE arithmetic.
// ---------------------------------------------------------------------------
KV_HD float recip(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
KV_HD double recip(double x) { return 1.0 / x; }

KV_HD float rsqrt_pos(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
KV_HD double rsqrt_pos(double x) { return 1.0 / sqrt(x); }

KV_HD float log_pos(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_logf(x) * 0.693147180559945309f;   // v_log_f32 is log2
#else
    return logf(x);
#endif
}
KV_HD double log_pos(double x) { return log(x); }

// ---------------------------------------------------------------------------
// Fixed-size dense helpers on row-major register arrays.
// ---------------------------------------------------------------------------

// C (M x N) {=, +=, -=} op(A) (M x Kd) * op(B) (Kd x N).
// TA: A is stored Kd x M and used transposed.  TB: B is stored N x Kd.
// MODE 0: assign, 1: add, 2: subtract.
template <typename R, int M, int Kd, int N, bool TA, bool TB, int MODE>
KV_HD void gemm(const R* A, const R* Bm, R* C) {
    KV_UNROLL
    for (int i = 0; i < M; ++i) {
        KV_UNROLL
        for (int j = 0; j < N; ++j) {
            R acc = R(0);
            KV_UNROLL
            for (int k = 0; k < Kd; ++k) {
                const R a = TA ? A[k * M + i] : A[i * Kd + k];
                const R b = TB ? Bm[j * Kd + k] : Bm[k * N + j];
                acc += a * b;
            }
            if (MODE == 0) C[i * N + j] = acc;
            if (MODE == 1) C[i * N + j] += acc;
            if (MODE == 2) C[i * N + j] -= acc;
        }
    }
}

// y (M) {=,+=,-=} op(A) x ; A is M x N, or N x M when TA.
template <typename R, int M, int N, bool TA, int MODE>
KV_HD void gemv(const R* A, const R* x, R* y) {
    KV_UNROLL
    for (int i = 0; i < M; ++i) {
        R acc = R(0);
        KV_UNROLL
        for (int j = 0; j < N; ++j) acc += (TA ? A[j * M + i] : A[i * N + j]) * x[j];
        if (MODE == 0) y[i] = acc;
        if (MODE == 1) y[i] += acc;
        if (MODE == 2) y[i] -= acc;
    }
}

// C (M x N) += s * x y^T
template <typename R, int M, int N>
KV_HD void outer_add(R s, const R* x, const R* y, R* C) {
    KV_UNROLL
    for (int i = 0; i < M; ++i) {
        KV_UNROLL
        for (int j = 0; j < N; ++j) C[i * N + j] += s * x[i] * y[j];
    }
}

template <typename R, int N>
KV_HD void symmetrize_inplace(R* X) {
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        KV_UNROLL
        for (int j = i + 1; j < N; ++j) {
            const R v = (X[i * N + j] + X[j * N + i]) / R(2);
            X[i * N + j] = v;
            X[j * N + i] = v;
        }
    }
}

template <typename R, int Len>
KV_HD void vzero(R* x) {
    KV_UNROLL
    for (int i = 0; i < Len; ++i) x[i] = R(0);
}

template <typename R, int Len>
KV_HD void vcopy(const R* s, R* d) {
    KV_UNROLL
    for (int i = 0; i < Len; ++i) d[i] = s[i];
}

template <typename R, int Len>
KV_HD void vadd(const R* s, R* d) {
    KV_UNROLL
    for (int i = 0; i < Len; ++i) d[i] += s[i];
}

template <typename R, int Len>
KV_HD R vdot(const R* x, const R* y) {
    R acc = R(0);
    KV_UNROLL
    for (int i = 0; i < Len; ++i) acc += x[i] * y[i];
    return acc;
}

// Global memory <-> registers.  `p` may be null for optional tensors.
template <typename R, int Len>
KV_HD void gload(const R* p, R* x) {
    KV_UNROLL
    for (int i = 0; i < Len; ++i) x[i] = p[i];
}
template <typename R, int Len>
KV_HD void gload_or_zero(const R* base, long off, R* x) {
    if (base) {
        gload<R, Len>(base + off, x);
    } else {
        vzero<R, Len>(x);
    }
}
template <typename R, int Len>
KV_HD void gstore(R* p, const R* x) {
    KV_UNROLL
    for (int i = 0; i < Len; ++i) p[i] = x[i];
}

// Lower Cholesky factor of a symmetric matrix (only its lower triangle is
// read, like LAPACK potrf('L')).  Returns false if a pivot was <= 0; the
// factor is then completed with that pivot replaced by 1 so nothing is NaN.
template <typename R, int N>
KV_HD bool cholesky(const R* S, R* L) {
    bool ok = true;
    KV_UNROLL
    for (int j = 0; j < N; ++j) {
        R d = S[j * N + j];
        KV_UNROLL
        for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k];
        if (!(d > R(0))) {
            ok = false;
            d = R(1);
        }
        const R inv = rsqrt_pos(d);
        L[j * N + j] = d * inv;
        KV_UNROLL
        for (int i = j + 1; i < N; ++i) {
            R s = S[i * N + j];
            KV_UNROLL
            for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k];
            L[i * N + j] = s * inv;
        }
        KV_UNROLL
        for (int i = 0; i < j; ++i) L[i * N + j] = R(0);
    }
    return ok;
}

// Inverse of a lower-triangular matrix (result lower-triangular).
template <typename R, int N>
KV_HD void tri_lower_inverse(const R* L, R* Li) {
    KV_UNROLL
    for (int i = 0; i < N; ++i) Li[i * N + i] = recip(L[i * N + i]);
    KV_UNROLL
    for (int j = 0; j < N; ++j) {
        KV_UNROLL
        for (int i = 0; i < j; ++i) Li[i * N + j] = R(0);
        KV_UNROLL
        for (int i = j + 1; i < N; ++i) {
            R s = R(0);
            KV_UNROLL
            for (int k = j; k < i; ++k) s += L[i * N + k] * Li[k * N + j];
            Li[i * N + j] = -s * Li[i * N + i];
        }
    }
}

// Inverse of a symmetric positive definite matrix through its Cholesky
// factor: S^-1 = L^-T L^-1.  Optionally returns L and L^-1.
template <typename R, int N>
KV_HD bool spd_inverse(const R* S, R* Si, R* L_out = nullptr, R* Li_out = nullptr) {
    R L[N * N], Li[N * N];
    const bool ok = cholesky<R, N>(S, L);
    tri_lower_inverse<R, N>(L, Li);
    gemm<R, N, N, N, true, false, 0>(Li, Li, Si);
    if (L_out) vcopy<R, N * N>(L, L_out);
    if (Li_out) vcopy<R, N * N>(Li, Li_out);
    return ok;
}

// x := L^-1 x (forward substitution), L lower-triangular.
template <typename R, int N>
KV_HD void solve_lower(const R* L, R* x) {
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        R s = x[i];
        KV_UNROLL
        for (int k = 0; k < i; ++k) s -= L[i * N + k] * x[k];
        x[i] = s * recip(L[i * N + i]);
    }
}
// x := L^-T x (back substitution).
template <typename R, int N>
KV_HD void solve_lower_t(const R* L, R* x) {
    KV_UNROLL
    for (int i = N - 1; i >= 0; --i) {
        R s = x[i];
        KV_UNROLL
        for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
        x[i] = s * recip(L[i * N + i]);
    }
}

// Adjoint of the Cholesky factorisation S = L L^T: given Lbar (only its lower
// triangle is meaningful) add the symmetric gradient
//   Sbar = sym( L^-T Phi(L^T Lbar) L^-1 ),  Phi = lower triangle, diagonal halved
// to Sacc.
template <typename R, int N>
KV_HD void cholesky_adjoint_add(const R* L, const R* Lbar, R* Sacc) {
    R Li[N * N], X[N * N], Y[N * N], Z[N * N];
    tri_lower_inverse<R, N>(L, Li);
    // X = Phi(L^T tril(Lbar))
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        KV_UNROLL
        for (int j = 0; j < N; ++j) {
            R acc = R(0);
            if (i >= j) {
                KV_UNROLL
                for (int k = i; k < N; ++k) acc += L[k * N + i] * Lbar[k * N + j];
                if (i == j) acc *= R(0.5);
            }
            X[i * N + j] = acc;
        }
    }
    gemm<R, N, N, N, true, false, 0>(Li, X, Y);   // Y = L^-T X
    gemm<R, N, N, N, false, false, 0>(Y, Li, Z);  // Z = Y L^-1
    KV_UNROLL
    for (int i = 0; i < N; ++i) {
        KV_UNROLL
        for (int j = 0; j < N; ++j) Sacc[i * N + j] += (Z[i * N + j] + Z[j * N + i]) * R(0.5);
    }
}

// out (r x c) = sum_k w[k] * bases[k] ; bases is (K, r*c) in global memory
// (identical for every lane, so the loads are wave-uniform).
template <typename R, int K, int Len>
KV_HD void mix(const R* w, const R* bases, R* out) {
    KV_UNROLL
    for (int e = 0; e < Len; ++e) {
        R acc = R(0);
        KV_UNROLL
        for (int k = 0; k < K; ++k) acc += w[k] * bases[k * Len + e];
        out[e] = acc;
    }
}

// ---------------------------------------------------------------------------
// Argument blocks (device pointers; see include/kvae_kalman.h for shapes).
// ---------------------------------------------------------------------------
template <typename R>
struct FilterFwdArgs {
    int T, B, symmetrize, carry_per_sequence;
    const R *a, *alpha, *mask, *A_K, *C_K, *Q, *Rm, *m_in, *P_in;
    R *m, *Pf, *mn, *Pn, *As, *Cs;
    int* status;
};

template <typename R>
struct FilterBwdArgs {
    FilterFwdArgs<R> f;
    int burn_in;
    const R *g_m, *g_P, *g_mn, *g_Pn, *g_As, *g_Cs;
    R *d_a, *d_alpha, *d_m_in, *d_P_in;
    R* partial;  // (rows, FilterParamGrads::LEN) per-wave partial sums
    // wavefront-per-sequence variant only: gradients w.r.t. the mixed per-step
    // matrices, (T+1,B,n,n) and (T+1,B,p,n); the mixing adjoint is a separate pass
    R *bar_As, *bar_Cs;
};

template <typename R>
struct SmoothFwdArgs {
    int T, B, symmetrize;
    const R *m, *Pf, *mn, *Pn, *As, *Cs, *LR, *eps_s, *eps_o;
    R *ms, *Ps, *zs, *ares;
    int* status;
};

template <typename R>
struct SmoothBwdArgs {
    SmoothFwdArgs<R> f;
    int burn_in;
    const R *g_ms, *g_Ps, *g_zs, *g_ares;
    R *d_m, *d_P, *d_mn, *d_Pn, *d_As, *d_Cs;
    R* partial;  // (rows, P*P) partial sums of d_LR, or null
};

template <typename R>
struct ElboFwdArgs {
    int T, B;
    const R *a, *ms, *Ps, *As, *Cs, *Q, *Rm, *m0, *P0, *eps;
    R *obs, *trans, *post, *z;
    int* status;
};

template <typename R>
struct ElboBwdArgs {
    ElboFwdArgs<R> f;
    const R *g_obs, *g_trans, *g_post;
    R *d_a, *d_ms, *d_Ps, *d_As, *d_Cs;
    R* partial;  // (rows, ElboParamGrads::LEN)
    // when set, the cotangent of every entry of a term is weight * g_scale[k * stride]
    // (the terms only feed an aggregated objective) and g_obs/g_trans/g_post are unused
    const R* g_scale = nullptr;
    int g_scale_stride = 0;
    R w_obs = R(0), w_trans = R(0), w_post = R(0);
};

// upstream gradients of the three terms at pair tb
template <typename R>
KV_HD void elbo_cotangents(const ElboBwdArgs<R>& g, long tb, R& wo, R& wt, R& wp) {
    if (g.g_scale) {
        wo = g.w_obs * g.g_scale[0];
        wt = g.w_trans * g.g_scale[g.g_scale_stride];
        wp = g.w_post * g.g_scale[2 * g.g_scale_stride];
    } else {
        wo = g.g_obs ? g.g_obs[tb] : R(0);
        wt = g.g_trans ? g.g_trans[tb] : R(0);
        wp = g.g_post ? g.g_post[tb] : R(0);
    }
}

template <int N, int P, int K>
struct FilterParamGrads {
    static constexpr int OFF_A = 0;
    static constexpr int OFF_C = OFF_A + K * N * N;
    static constexpr int OFF_Q = OFF_C + K * P * N;
    static constexpr int OFF_R = OFF_Q + N * N;
    static constexpr int LEN = OFF_R + P * P;
};

template <int N, int P>
struct ElboParamGrads {
    static constexpr int OFF_Q = 0;
    static constexpr int OFF_R = OFF_Q + N * N;
    static constexpr int OFF_M0 = OFF_R + P * P;
    static constexpr int OFF_P0 = OFF_M0 + N;
    static constexpr int LEN = OFF_P0 + N * N;
};

// ---------------------------------------------------------------------------
// Filter, forward.  state_space_model.py:284-352.
// ---------------------------------------------------------------------------

// Quantities of one measurement update that forward and backward both need.
template <typename R, int N, int P>
struct UpdateTerms {
    R M[P * N];   // C Pp
    R Si[P * P];  // (C Pp C^T + R)^-1
    R F[N * P];   // Pp C^T
    R G[N * P];   // gain  F Si
    R r[P];       // innovation a - C mp
};

template <typename R, int N, int P>
KV_HD bool update_terms(const R* C, const R* mp, const R* Pp, const R* Rm, const R* a,
                        UpdateTerms<R, N, P>& u) {
    R S[P * P];
    gemm<R, P, N, N, false, false, 0>(C, Pp, u.M);
    vcopy<R, P * P>(Rm, S);
    gemm<R, P, N, P, false, true, 1>(u.M, C, S);  // S = M C^T + R
    const bool ok = spd_inverse<R, P>(S, u.Si);
    gemm<R, N, N, P, false, true, 0>(Pp, C, u.F);
    gemm<R, N, P, P, false, false, 0>(u.F, u.Si, u.G);
    vcopy<R, P>(a, u.r);
    gemv<R, P, N, false, 2>(C, mp, u.r);
    return ok;
}

// `next_weights(t, tb, w, C, mp, a, wn)` supplies the mixing weights of step t+1:
// read from g.alpha by filter_forward_sequence, computed in the loop when the weight
// network is evaluated in the kernel (masked sequences, state_space_model.py:232-282).
template <typename R, int N, int P, int K, typename Next>
KV_HD void filter_forward_loop(const FilterFwdArgs<R>& g, int b, Next&& next_weights) {
    const long B = g.B;
    R Rm[P * P], Q[N * N];
    gload<R, P * P>(g.Rm, Rm);
    gload<R, N * N>(g.Q, Q);

    R mp[N], Pp[N * N];
    if (g.carry_per_sequence) {
        gload<R, N>(g.m_in + (long)b * N, mp);
        gload<R, N * N>(g.P_in + (long)b * N * N, Pp);
    } else {
        gload<R, N>(g.m_in, mp);
        gload<R, N * N>(g.P_in, Pp);
    }

    R w[K], wn[K];
    gload<R, K>(g.alpha + (long)b * K, w);
    if (g.As) {
        R A0[N * N];
        mix<R, K, N * N>(w, g.A_K, A0);
        gstore<R, N * N>(g.As + (long)b * N * N, A0);
    }
    int st = 0;
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        R C[P * N], An[N * N], a[P];
        gload<R, P>(g.a + tb * P, a);
        mix<R, K, P * N>(w, g.C_K, C);
        next_weights(t, tb, w, C, mp, a, wn);
        mix<R, K, N * N>(wn, g.A_K, An);
        if (g.Cs) gstore<R, P * N>(g.Cs + tb * P * N, C);
        if (g.As) gstore<R, N * N>(g.As + (tb + B) * N * N, An);

        UpdateTerms<R, N, P> u;
        if (!update_terms<R, N, P>(C, mp, Pp, Rm, a, u)) st |= ST_NONPD_INNOVATION;

        R m[N], Pf[N * N];
        vcopy<R, N>(mp, m);
        gemv<R, N, P, false, 1>(u.G, u.r, m);  // m_obs = mp + G r
        vcopy<R, N * N>(Pp, Pf);
        gemm<R, N, P, N, false, false, 2>(u.G, u.M, Pf);  // P_obs = Pp - G (C Pp)
        if (g.mask) {
            const R mu = g.mask[tb];
            const R nu = R(1) - mu;
            KV_UNROLL
            for (int i = 0; i < N; ++i) m[i] = mu * m[i] + nu * mp[i];
            KV_UNROLL
            for (int i = 0; i < N * N; ++i) Pf[i] = mu * Pf[i] + nu * Pp[i];
        }
        if (g.symmetrize) symmetrize_inplace<R, N>(Pf);

        R T1[N * N];
        gemv<R, N, N, false, 0>(An, m, mp);                 // m_{t+1|t}
        gemm<R, N, N, N, false, false, 0>(An, Pf, T1);      // An P
        vcopy<R, N * N>(Q, Pp);
        gemm<R, N, N, N, false, true, 1>(T1, An, Pp);       // An P An^T + Q
        if (g.symmetrize) symmetrize_inplace<R, N>(Pp);

        gstore<R, N>(g.m + tb * N, m);
        gstore<R, N * N>(g.Pf + tb * N * N, Pf);
        if (g.mn) gstore<R, N>(g.mn + tb * N, mp);          // null: the one-pass filter +
        if (g.Pn) gstore<R, N * N>(g.Pn + tb * N * N, Pp);  // smoother recomputes them
        vcopy<R, K>(wn, w);
    }
    if (g.Cs) {
        R C[P * N];
        mix<R, K, P * N>(w, g.C_K, C);
        gstore<R, P * N>(g.Cs + ((long)g.T * B + b) * P * N, C);
    }
    if (st && g.status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(g.status, st);
#else
        *g.status |= st;
#endif
    }
}

template <typename R, int N, int P, int K>
KV_HD void filter_forward_sequence(const FilterFwdArgs<R>& g, int b) {
    const long B = g.B;
    filter_forward_loop<R, N, P, K>(
        g, b, [&](int, long tb, const R*, const R*, const R*, const R*, R* wn) {
            gload<R, K>(g.alpha + (tb + B) * K, wn);
        });
}

// The weight network of masked sequences evaluated inside the filter loop, for the
// two-layer perceptron  softmax(W2 relu(W1 a + b1) + b2)
// (dynamics_parameter_network.py:61-90 with num_layers = 2).  Per step, before the
// measurement update (state_space_model.py:232-282):
//   z = A(w) m_pred (+ L_Q eps_z),  a_imp = C(w) z (+ L_R eps_a),
//   a_w = mu a + (1 - mu) a_imp,    w_next = net(a_w)
// `net` is the packed network: per hidden unit j the P entries of W1[j], b1[j] and
// the K entries of W2[:, j]; then b2.  a_w (T,B,P) and rows 1..T of alpha_out
// (T+1,B,K) are written; row 0 of alpha_out holds the weights of the first step.
// softmax(W2 relu(W1 a + b1) + b2) from the packed network: per hidden unit j the P
// entries of W1[j], b1[j] and the K entries of W2[:, j]; then b2
// (dynamics_parameter_network.py:61-90 with num_layers = 2).
template <typename R, int P, int K>
KV_HD void mlp2_softmax(const R* net, int hidden, const R* a, R* w) {
    constexpr int UNIT = P + 1 + K;
    R logit[K];
    KV_UNROLL
    for (int k = 0; k < K; ++k) logit[k] = net[(long)hidden * UNIT + k];
#pragma GCC unroll 8
    for (int j = 0; j < hidden; ++j) {   // uniform loads: several units in flight per wait
        const R* u = net + (long)j * UNIT;
        R h = u[P];
        KV_UNROLL
        for (int i = 0; i < P; ++i) h += u[i] * a[i];
        h = h > R(0) ? h : R(0);
        KV_UNROLL
        for (int k = 0; k < K; ++k) logit[k] += u[P + 1 + k] * h;
    }
    R top = logit[0];
    KV_UNROLL
    for (int k = 1; k < K; ++k) top = logit[k] > top ? logit[k] : top;
    R total = R(0);
    KV_UNROLL
    for (int k = 0; k < K; ++k) {
        w[k] = exp(logit[k] - top);
        total += w[k];
    }
    const R inv_total = recip(total);
    KV_UNROLL
    for (int k = 0; k < K; ++k) w[k] *= inv_total;
}

template <typename R>
struct MaskedNetArgs {
    int hidden;
    const R* net;       // global or LDS copy of the packed network
    const R *eps_z, *L_Q, *eps_a, *L_R;   // (T,B,N), (N,N), (T,B,P), (P,P); pairs may be null
    R *a_w, *alpha_out;
};

template <typename R, int N, int P, int K>
KV_HD void filter_forward_masked_mlp_sequence(const FilterFwdArgs<R>& g,
                                              const MaskedNetArgs<R>& q, int b) {
    const long B = g.B;
    filter_forward_loop<R, N, P, K>(
        g, b, [&](int, long tb, const R* w, const R* C, const R* mp, const R* a, R* wn) {
            R A[N * N], z[N], aw[P];
            mix<R, K, N * N>(w, g.A_K, A);
            gemv<R, N, N, false, 0>(A, mp, z);
            if (q.eps_z) {
                R e[N], L[N * N];
                gload<R, N>(q.eps_z + tb * N, e);
                gload<R, N * N>(q.L_Q, L);
                gemv<R, N, N, false, 1>(L, e, z);
            }
            gemv<R, P, N, false, 0>(C, z, aw);
            if (q.eps_a) {
                R e[P], L[P * P];
                gload<R, P>(q.eps_a + tb * P, e);
                gload<R, P * P>(q.L_R, L);
                gemv<R, P, P, false, 1>(L, e, aw);
            }
            const R mu = g.mask ? g.mask[tb] : R(1);
            KV_UNROLL
            for (int i = 0; i < P; ++i) aw[i] = mu * a[i] + (R(1) - mu) * aw[i];
            gstore<R, P>(q.a_w + tb * P, aw);
            mlp2_softmax<R, P, K>(q.net, q.hidden, aw, wn);
            gstore<R, K>(q.alpha_out + (tb + B) * K, wn);
        });
}

// ---------------------------------------------------------------------------
// Filter, backward (reverse in time).
//
// Per step t, with carry (mpb, Ppb) = gradient w.r.t. the prediction
// (m_{t+1|t}, P_{t+1|t}) that the NEXT step consumed:
//   1. add the upstream gradients of the four outputs of step t; zero all
//      four when t < burn_in (the reference detaches outputs and carry alike,
//      state_space_model.py:342-346);
//   2. adjoint of the time update  m' = An m, P' = sym(An P An^T + Q);
//   3. adjoint of symmetrise, mask blend, covariance and mean update, gain,
//      innovation covariance and its inverse;
//   4. scatter Cbar_t / Abar_{t+1} into d_alpha and the basis gradients.
// acc holds this lane's running sums of the parameter gradients.
// ---------------------------------------------------------------------------
// ACCUM: d_a (T,B,P) and rows 0..T-1 of d_alpha (T+1,B,K) already hold a first part of
// the gradient (written by the caller's earlier pass over the same sequence) and this
// pass adds to them; row T of d_alpha is written.  The gradient w.r.t. the initial
// (m_in, P_in) of the sequence is returned in m_in_bar (N) / P_in_bar (N*N).
// Net: NoNetAdjoint, or MlpNetAdjoint when the weights of step t+1 were produced inside the
// filter loop by the perceptron network from the imputed observation of step t
// (filter_forward_masked_mlp_sequence); the gradient of the weights then flows on through
// the network into the observation, the imputation and the prediction it read.
struct NoNetAdjoint {
    static constexpr bool active = false;
};

template <typename R, int N, int P, int K, bool ACCUM, typename Net = NoNetAdjoint>
KV_HD void filter_backward_core(const FilterBwdArgs<R>& g, int b, R* acc, R* m_in_bar,
                                R* P_in_bar, const Net& net = Net()) {
    using PG = FilterParamGrads<N, P, K>;
    const FilterFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    R Rm[P * P];
    gload<R, P * P>(f.Rm, Rm);
    R At_bar[N * N];   // Net: gradient w.r.t. A_t from the imputation of step t (A_t = An of t-1)
    vzero<R, N * N>(At_bar);

    R mpb[N], Ppb[N * N];  // carry: grad wrt prediction made at step t
    vzero<R, N>(mpb);
    vzero<R, N * N>(Ppb);
    R dwn[K];  // C-part of d_alpha[t+1], produced by step t+1
    {
        // d_alpha[T] starts with <g_Cs[T], C^k>; C_T's basis gradient.
        R gC[P * N], wT[K];
        gload_or_zero<R, P * N>(g.g_Cs, ((long)T * B + b) * P * N, gC);
        gload<R, K>(f.alpha + ((long)T * B + b) * K, wT);
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            R s = R(0);
            KV_UNROLL
            for (int e = 0; e < P * N; ++e) {
                s += gC[e] * f.C_K[k * P * N + e];
                acc[PG::OFF_C + k * P * N + e] += wT[k] * gC[e];
            }
            dwn[k] = s;
        }
    }

    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        R w[K], wn[K], C[P * N], An[N * N], a[P];
        gload<R, K>(f.alpha + tb * K, w);
        gload<R, K>(f.alpha + (tb + B) * K, wn);
        gload<R, P>(f.a + tb * P, a);
        mix<R, K, P * N>(w, f.C_K, C);
        mix<R, K, N * N>(wn, f.A_K, An);

        // forward values of this step
        R mp[N], Pp[N * N], m[N], Pf[N * N];
        if (t > 0) {
            gload<R, N>(f.mn + (tb - B) * N, mp);
            gload<R, N * N>(f.Pn + (tb - B) * N * N, Pp);
        } else if (f.carry_per_sequence) {
            gload<R, N>(f.m_in + (long)b * N, mp);
            gload<R, N * N>(f.P_in + (long)b * N * N, Pp);
        } else {
            gload<R, N>(f.m_in, mp);
            gload<R, N * N>(f.P_in, Pp);
        }
        gload<R, N>(f.m + tb * N, m);
        gload<R, N * N>(f.Pf + tb * N * N, Pf);

        // 1. gather gradients of the step's outputs
        R mb[N], Pb[N * N];
        gload_or_zero<R, N>(g.g_m, tb * N, mb);
        gload_or_zero<R, N * N>(g.g_P, tb * N * N, Pb);
        if (g.g_mn) {
            KV_UNROLL
            for (int i = 0; i < N; ++i) mpb[i] += g.g_mn[tb * N + i];
        }
        if (g.g_Pn) {
            KV_UNROLL
            for (int i = 0; i < N * N; ++i) Ppb[i] += g.g_Pn[tb * N * N + i];
        }
        if (t < g.burn_in) {
            vzero<R, N>(mb);
            vzero<R, N * N>(Pb);
            vzero<R, N>(mpb);
            vzero<R, N * N>(Ppb);
        }

        // 2. time update adjoint
        R Ab[N * N];
        gload_or_zero<R, N * N>(g.g_As, (tb + B) * N * N, Ab);
        {
            if (f.symmetrize) symmetrize_inplace<R, N>(Ppb);  // H
            R T1[N * N], T2[N * N];
            // Abar += mpb m^T + H An Pf^T + H^T An Pf
            outer_add<R, N, N>(R(1), mpb, m, Ab);
            gemm<R, N, N, N, false, true, 0>(An, Pf, T1);   // An Pf^T
            gemm<R, N, N, N, false, false, 1>(Ppb, T1, Ab);
            gemm<R, N, N, N, false, false, 0>(An, Pf, T1);  // An Pf
            gemm<R, N, N, N, true, false, 1>(Ppb, T1, Ab);
            // mbar += An^T mpb ; Pbar += An^T H An ; Qbar += H
            gemv<R, N, N, true, 1>(An, mpb, mb);
            gemm<R, N, N, N, true, false, 0>(An, Ppb, T2);  // An^T H
            gemm<R, N, N, N, false, false, 1>(T2, An, Pb);
            vadd<R, N * N>(Ppb, acc + PG::OFF_Q);
        }
        const R mu = f.mask ? f.mask[tb] : R(1);
        const R nu = R(1) - mu;
        R Cb[P * N];
        gload_or_zero<R, P * N>(g.g_Cs, tb * P * N, Cb);

        // 2b. A-part of the mixing adjoint: d_alpha[t+1] is complete here
        R da[K];
        {
            if constexpr (Net::active) vadd<R, N * N>(At_bar, Ab);   // imputation of step t+1
            KV_UNROLL
            for (int k = 0; k < K; ++k) {
                R sA = R(0);
                KV_UNROLL
                for (int e = 0; e < N * N; ++e) {
                    sA += Ab[e] * f.A_K[k * N * N + e];
                    acc[PG::OFF_A + k * N * N + e] += wn[k] * Ab[e];
                }
                da[k] = dwn[k] + sA;
            }
            if (ACCUM && t + 1 < T) {
                KV_UNROLL
                for (int k = 0; k < K; ++k) da[k] += g.d_alpha[(tb + B) * K + k];
            }
        }
        // 2c. through the network that produced alpha[t+1] from the imputed observation
        R a_extra[P], mp_extra[N];
        if constexpr (Net::active) {
            net.template step<N, P, K>(f, t, tb, da, w, wn, C, mp, mu, a_extra, Cb, At_bar, mp_extra);
        }
        gstore<R, K>(g.d_alpha + (tb + B) * K, da);

        // 3. measurement update adjoint
        UpdateTerms<R, N, P> u;
        update_terms<R, N, P>(C, mp, Pp, Rm, a, u);
        if (f.symmetrize) symmetrize_inplace<R, N>(Pb);  // H
        // new carry: gradients w.r.t. (mp, Pp)
        R mob[N], Pob[N * N];  // grads of m_obs, P_obs
        KV_UNROLL
        for (int i = 0; i < N; ++i) {
            mob[i] = mu * mb[i];
            mpb[i] = nu * mb[i] + mob[i];  // blend + identity term of m_obs
        }
        KV_UNROLL
        for (int i = 0; i < N * N; ++i) {
            Pob[i] = mu * Pb[i];
            Ppb[i] = nu * Pb[i] + Pob[i];  // blend + identity term of P_obs
        }
        R Gb[N * P];
        vzero<R, N * P>(Gb);
        // m_obs = mp + G r ; r = a - C mp
        outer_add<R, N, P>(R(1), mob, u.r, Gb);
        R rb[P];
        gemv<R, P, N, true, 0>(u.G, mob, rb);  // G^T mob
        if (ACCUM) {
            R da_t[P];
            gload<R, P>(g.d_a + tb * P, da_t);
            KV_UNROLL
            for (int i = 0; i < P; ++i) da_t[i] += rb[i];
            gstore<R, P>(g.d_a + tb * P, da_t);
        } else if constexpr (Net::active) {
            R da_t[P];
            KV_UNROLL
            for (int i = 0; i < P; ++i) da_t[i] = rb[i] + a_extra[i];
            gstore<R, P>(g.d_a + tb * P, da_t);
        } else {
            gstore<R, P>(g.d_a + tb * P, rb);
        }
        outer_add<R, P, N>(R(-1), rb, mp, Cb);
        gemv<R, N, P, true, 2>(C, rb, mpb);  // mpb -= C^T rb
        // P_obs = Pp - G M ; M = C Pp
        gemm<R, N, N, P, false, true, 2>(Pob, u.M, Gb);  // Gb -= Pob M^T
        R Mb[P * N];
        gemm<R, P, N, N, true, false, 0>(u.G, Pob, Mb);  // G^T Pob  (sign below)
        gemm<R, P, N, N, false, true, 2>(Mb, Pp, Cb);    // Cb += (-G^T Pob) Pp^T
        gemm<R, N, P, N, true, false, 2>(C, Mb, Ppb);    // Ppb += C^T (-G^T Pob)
        // G = F Si ; F = Pp C^T
        R Fb[N * P], Sib[P * P], Sb[P * P], T3[P * P];
        gemm<R, N, P, P, false, true, 0>(Gb, u.Si, Fb);       // Gb Si^T
        gemm<R, P, N, P, true, false, 0>(u.F, Gb, Sib);       // F^T Gb
        gemm<R, P, P, P, true, false, 0>(u.Si, Sib, T3);      // Si^T Sib
        gemm<R, P, P, P, false, true, 0>(T3, u.Si, Sb);       // ... Si^T
        KV_UNROLL
        for (int i = 0; i < P * P; ++i) Sb[i] = -Sb[i];
        gemm<R, N, P, N, false, false, 1>(Fb, C, Ppb);        // Ppb += Fb C
        gemm<R, P, N, N, true, false, 1>(Fb, Pp, Cb);         // Cb += Fb^T Pp
        // S = C Pp C^T + R
        {
            R CPt[P * N];  // C Pp^T
            gemm<R, P, N, N, false, true, 0>(C, Pp, CPt);
            gemm<R, P, P, N, false, false, 1>(Sb, CPt, Cb);   // Sb C Pp^T
            gemm<R, P, P, N, true, false, 1>(Sb, u.M, Cb);    // Sb^T C Pp
            R T4[P * N];
            gemm<R, P, P, N, false, false, 0>(Sb, C, T4);     // Sb C
            gemm<R, N, P, N, true, false, 1>(C, T4, Ppb);     // C^T Sb C
            vadd<R, P * P>(Sb, acc + PG::OFF_R);
        }

        // 4. C-part of the mixing adjoint (the C-part of d_alpha[t])
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            R sC = R(0);
            KV_UNROLL
            for (int e = 0; e < P * N; ++e) {
                sC += Cb[e] * f.C_K[k * P * N + e];
                acc[PG::OFF_C + k * P * N + e] += w[k] * Cb[e];
            }
            dwn[k] = sC;
        }
        if constexpr (Net::active) vadd<R, N>(mp_extra, mpb);   // the imputation read mp too
    }
    // d_alpha[0]: C-part of step 0 plus <g_As[0], A^k>; A_0's basis gradient.
    {
        R gA[N * N], w0[K], da[K];
        gload_or_zero<R, N * N>(g.g_As, (long)b * N * N, gA);
        if constexpr (Net::active) vadd<R, N * N>(At_bar, gA);   // imputation of step 0
        gload<R, K>(f.alpha + (long)b * K, w0);
        KV_UNROLL
        for (int k = 0; k < K; ++k) {
            R s = R(0);
            KV_UNROLL
            for (int e = 0; e < N * N; ++e) {
                s += gA[e] * f.A_K[k * N * N + e];
                acc[PG::OFF_A + k * N * N + e] += w0[k] * gA[e];
            }
            da[k] = dwn[k] + s;
            if (ACCUM) da[k] += g.d_alpha[(long)b * K + k];
        }
        gstore<R, K>(g.d_alpha + (long)b * K, da);
    }
    vcopy<R, N>(mpb, m_in_bar);
    vcopy<R, N * N>(Ppb, P_in_bar);
}

template <typename R, int N, int P, int K>
KV_HD void filter_backward_sequence(const FilterBwdArgs<R>& g, int b, R* acc) {
    R mpb[N], Ppb[N * N];
    filter_backward_core<R, N, P, K, false>(g, b, acc, mpb, Ppb);
    gstore<R, N>(g.d_m_in + (long)b * N, mpb);
    gstore<R, N * N>(g.d_P_in + (long)b * N * N, Ppb);
}

// Adjoint of the in-loop perceptron and of the imputation that feeds it
// (filter_forward_masked_mlp_sequence; state_space_model.py:232-282), per step t once the
// gradient `da` of alpha[t+1] is complete:
//   softmax:     dl_k = wn_k (da_k - sum_j da_j wn_j)                      -> d_logits[t]
//   perceptron:  d a_w = sum_j W1[j,:] [pre_j > 0] sum_k W2[k,j] dl_k      (+ upstream g_a_w)
//   blend:       d a_t += mu d a_w;   d a_imp = (1 - mu) d a_w
//   imputation:  a_imp = C_t z (+ L_R e_a), z = A_t mp (+ L_Q e_z):
//                Cbar_t += d a_imp z^T, d z = C_t^T d a_imp, Abar_t += d z mp^T,
//                d mp += A_t^T d z, d L_R += d a_imp e_a^T, d L_Q += d z e_z^T
// The gradients of the network's own parameters are left to the caller, who re-evaluates
// the network on a_w and back-propagates d_logits (two small matrix products).
template <typename R>
struct MlpNetAdjoint {
    static constexpr bool active = true;
    MaskedNetArgs<R> q;        // forward arguments; q.a_w is read
    const R *g_a_w, *g_alpha;  // upstream gradients (T,B,P) / (T+1,B,K), may be null
    R* d_logits;               // (T,B,K)
    R *acc_LQ, *acc_LR;        // this lane's running sums (N*N / P*P), may be null
    int learn;                 // 0: the weights were detached from the network's output
                               // (learn_weight_model = False, :279-282): only g_a_w flows

    template <int N, int P, int K>
    KV_HD void step(const FilterFwdArgs<R>& f, int t, long tb, R* da, const R* w, const R* wn,
                    const R* C, const R* mp, R mu, R* a_extra, R* Cb, R* At_bar,
                    R* mp_extra) const {
        constexpr int UNIT = P + 1 + K;
        const long B = f.B;
        if (g_alpha) {
            KV_UNROLL
            for (int k = 0; k < K; ++k) da[k] += g_alpha[(tb + B) * K + k];
        }
        R dl[K];
        {
            R dot = R(0);
            KV_UNROLL
            for (int k = 0; k < K; ++k) dot += da[k] * wn[k];
            KV_UNROLL
            for (int k = 0; k < K; ++k) dl[k] = learn ? wn[k] * (da[k] - dot) : R(0);
        }
        gstore<R, K>(d_logits + tb * K, dl);
        R aw[P], awb[P];
        gload<R, P>(q.a_w + tb * P, aw);
        gload_or_zero<R, P>(g_a_w, tb * P, awb);
#pragma GCC unroll 8
        for (int j = 0; j < q.hidden; ++j) {
            const R* u = q.net + (long)j * UNIT;
            R pre = u[P];
            KV_UNROLL
            for (int i = 0; i < P; ++i) pre += u[i] * aw[i];
            R hb = R(0);
            KV_UNROLL
            for (int k = 0; k < K; ++k) hb += u[P + 1 + k] * dl[k];
            hb = pre > R(0) ? hb : R(0);
            KV_UNROLL
            for (int i = 0; i < P; ++i) awb[i] += u[i] * hb;
        }
        R aib[P];
        KV_UNROLL
        for (int i = 0; i < P; ++i) {
            a_extra[i] = mu * awb[i];
            aib[i] = (R(1) - mu) * awb[i];
        }
        R A[N * N], z[N], zb[N];
        mix<R, K, N * N>(w, f.A_K, A);
        gemv<R, N, N, false, 0>(A, mp, z);
        if (q.eps_z) {
            R e[N], L[N * N];
            gload<R, N>(q.eps_z + tb * N, e);
            gload<R, N * N>(q.L_Q, L);
            gemv<R, N, N, false, 1>(L, e, z);
        }
        outer_add<R, P, N>(R(1), aib, z, Cb);
        gemv<R, N, P, true, 0>(C, aib, zb);          // C^T d a_imp
        vzero<R, N * N>(At_bar);
        outer_add<R, N, N>(R(1), zb, mp, At_bar);
        gemv<R, N, N, true, 0>(A, zb, mp_extra);     // A^T d z
        if (q.eps_a && acc_LR) {
            R e[P];
            gload<R, P>(q.eps_a + tb * P, e);
            outer_add<R, P, P>(R(1), aib, e, acc_LR);
        }
        if (q.eps_z && acc_LQ) {
            R e[N];
            gload<R, N>(q.eps_z + tb * N, e);
            outer_add<R, N, N>(R(1), zb, e, acc_LQ);
        }
    }
};

// acc: FilterParamGrads<N,P,K>::LEN running sums followed by N*N (d L_Q) and P*P (d L_R)
template <typename R, int N, int P, int K>
KV_HD void filter_backward_masked_mlp_sequence(const FilterBwdArgs<R>& g,
                                               const MlpNetAdjoint<R>& net, int b, R* acc) {
    R mpb[N], Ppb[N * N];
    filter_backward_core<R, N, P, K, false, MlpNetAdjoint<R>>(g, b, acc, mpb, Ppb, net);
    gstore<R, N>(g.d_m_in + (long)b * N, mpb);
    gstore<R, N * N>(g.d_P_in + (long)b * N * N, Ppb);
}

// ---------------------------------------------------------------------------
// Smoother, forward (reverse in time).  state_space_model.py:390-477.
// ---------------------------------------------------------------------------

// z = mean (+ chol(cov) eps), a_re = C z (+ LR eps').  Returns false when the
// covariance was not positive definite.
template <typename R, int N, int P>
KV_HD bool smooth_emit(const SmoothFwdArgs<R>& g, long tb, const R* mean, const R* cov,
                       const R* C, const R* LR) {
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
    gstore<R, N>(g.zs + tb * N, z);
    gstore<R, P>(g.ares + tb * P, ar);
    return ok;
}

template <typename R, int N, int P>
KV_HD void smooth_forward_sequence(const SmoothFwdArgs<R>& g, int b) {
    const long B = g.B;
    const int T = g.T;
    R LR[P * P];
    if (g.eps_o) {
        gload<R, P * P>(g.LR, LR);
    } else {
        vzero<R, P * P>(LR);
    }
    int st = 0;
    R ms[N], Ps[N * N], C[P * N];
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(g.m + tb * N, ms);
        gload<R, N * N>(g.Pf + tb * N * N, Ps);
        gload<R, P * N>(g.Cs + (tb + B) * P * N, C);  // C_T, the extra entry
        gstore<R, N>(g.ms + tb * N, ms);
        gstore<R, N * N>(g.Ps + tb * N * N, Ps);
        if (!smooth_emit<R, N, P>(g, tb, ms, Ps, C, LR)) st |= ST_NONPD_SMOOTHED;
    }
    for (int t = T - 2; t >= 0; --t) {
        const long tb = (long)t * B + b;
        R m[N], Pf[N * N], mn[N], Pn[N * N], An[N * N];
        gload<R, N>(g.m + tb * N, m);
        gload<R, N * N>(g.Pf + tb * N * N, Pf);
        gload<R, N>(g.mn + tb * N, mn);
        gload<R, N * N>(g.Pn + tb * N * N, Pn);
        gload<R, N * N>(g.As + (tb + B) * N * N, An);
        gload<R, P * N>(g.Cs + tb * P * N, C);

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
        if (!smooth_emit<R, N, P>(g, tb, ms, Ps, C, LR)) st |= ST_NONPD_SMOOTHED;
    }
    if (st && g.status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(g.status, st);
#else
        *g.status |= st;
#endif
    }
}

// ---------------------------------------------------------------------------
// Smoother, backward (forward in time).
//
// Carry (msb, Psb): gradient w.r.t. the smoothed (mean, cov) of step t,
// accumulated from step t-1's recursion.  Per step: add upstream gradients and
// the adjoint of the emission (z, a_re); zero when t < burn_in (the reference
// detaches before emitting, state_space_model.py:444-448); then the adjoint of
//   J = Pf An^T Pi, ms = m + J d, Ps = sym(Pf + J D J^T).
// acc_LR: running sum of the gradient of LR (P*P), may be null.
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
KV_HD void smooth_emit_adjoint(const SmoothBwdArgs<R>& g, long tb, const R* mean, const R* cov,
                               const R* C, R* msb, R* Psb, R* Cb, R* acc_LR) {
    const SmoothFwdArgs<R>& f = g.f;
    R zb[N], ab[P];
    gload_or_zero<R, N>(g.g_zs, tb * N, zb);
    gload_or_zero<R, P>(g.g_ares, tb * P, ab);
    vzero<R, P * N>(Cb);
    if (g.g_ares) {
        // a_re = C z + LR eps'
        R z[N];
        vcopy<R, N>(mean, z);
        R L[N * N], e[N];
        if (f.eps_s) {
            cholesky<R, N>(cov, L);
            gload<R, N>(f.eps_s + tb * N, e);
            gemv<R, N, N, false, 1>(L, e, z);
        }
        outer_add<R, P, N>(R(1), ab, z, Cb);
        gemv<R, N, P, true, 1>(C, ab, zb);
        if (f.eps_o && acc_LR) {
            R eo[P];
            gload<R, P>(f.eps_o + tb * P, eo);
            outer_add<R, P, P>(R(1), ab, eo, acc_LR);
        }
    }
    if (g.g_ares || g.g_zs) {
        vadd<R, N>(zb, msb);
        if (f.eps_s) {
            R L[N * N], e[N], Lb[N * N];
            cholesky<R, N>(cov, L);
            gload<R, N>(f.eps_s + tb * N, e);
            vzero<R, N * N>(Lb);
            outer_add<R, N, N>(R(1), zb, e, Lb);
            cholesky_adjoint_add<R, N>(L, Lb, Psb);
        }
    }
}

template <typename R, int N, int P>
KV_HD void smooth_backward_sequence(const SmoothBwdArgs<R>& g, int b, R* acc_LR) {
    const SmoothFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    R msb[N], Psb[N * N];
    vzero<R, N>(msb);
    vzero<R, N * N>(Psb);
    {
        // entries the recursion never touches
        R z1[N * N], z2[P * N];
        vzero<R, N * N>(z1);
        vzero<R, P * N>(z2);
        gstore<R, N * N>(g.d_As + (long)b * N * N, z1);                    // A_0
        gstore<R, N * N>(g.d_As + ((long)T * B + b) * N * N, z1);          // A_T
        gstore<R, P * N>(g.d_Cs + ((long)(T - 1) * B + b) * P * N, z2);    // C_{T-1}
        gstore<R, N>(g.d_mn + ((long)(T - 1) * B + b) * N, z1);
        gstore<R, N * N>(g.d_Pn + ((long)(T - 1) * B + b) * N * N, z1);
    }
    for (int t = 0; t < T - 1; ++t) {
        const long tb = (long)t * B + b;
        R m[N], Pf[N * N], mn[N], Pn[N * N], An[N * N], C[P * N];
        R ms1[N], Ps1[N * N], ms0[N], Ps0[N * N];
        gload<R, N>(f.m + tb * N, m);
        gload<R, N * N>(f.Pf + tb * N * N, Pf);
        gload<R, N>(f.mn + tb * N, mn);
        gload<R, N * N>(f.Pn + tb * N * N, Pn);
        gload<R, N * N>(f.As + (tb + B) * N * N, An);
        gload<R, P * N>(f.Cs + tb * P * N, C);
        gload<R, N>(f.ms + (tb + B) * N, ms1);
        gload<R, N * N>(f.Ps + (tb + B) * N * N, Ps1);
        gload<R, N>(f.ms + tb * N, ms0);
        gload<R, N * N>(f.Ps + tb * N * N, Ps0);

        // upstream + emission adjoint
        if (g.g_ms) {
            KV_UNROLL
            for (int i = 0; i < N; ++i) msb[i] += g.g_ms[tb * N + i];
        }
        if (g.g_Ps) {
            KV_UNROLL
            for (int i = 0; i < N * N; ++i) Psb[i] += g.g_Ps[tb * N * N + i];
        }
        R Cb[P * N];
        smooth_emit_adjoint<R, N, P>(g, tb, ms0, Ps0, C, msb, Psb, Cb, acc_LR);
        gstore<R, P * N>(g.d_Cs + tb * P * N, Cb);
        if (t < g.burn_in) {
            vzero<R, N>(msb);
            vzero<R, N * N>(Psb);
        }

        // recompute the step
        R Pi[N * N], W[N * N], J[N * N], d[N], D[N * N];
        spd_inverse<R, N>(Pn, Pi);
        gemm<R, N, N, N, true, false, 0>(An, Pi, W);
        gemm<R, N, N, N, false, false, 0>(Pf, W, J);
        KV_UNROLL
        for (int i = 0; i < N; ++i) d[i] = ms1[i] - mn[i];
        KV_UNROLL
        for (int i = 0; i < N * N; ++i) D[i] = Ps1[i] - Pn[i];

        if (f.symmetrize) symmetrize_inplace<R, N>(Psb);  // H
        // direct terms
        gstore<R, N>(g.d_m + tb * N, msb);
        R Pfb[N * N];
        vcopy<R, N * N>(Psb, Pfb);
        // Jbar = msb d^T + H J D^T + H^T J D
        R Jb[N * N], T1[N * N];
        vzero<R, N * N>(Jb);
        outer_add<R, N, N>(R(1), msb, d, Jb);
        gemm<R, N, N, N, false, true, 0>(J, D, T1);        // J D^T
        gemm<R, N, N, N, false, false, 1>(Psb, T1, Jb);
        gemm<R, N, N, N, false, false, 0>(J, D, T1);       // J D
        gemm<R, N, N, N, true, false, 1>(Psb, T1, Jb);
        // dbar = J^T msb ; Dbar = J^T H J
        R db[N], Db[N * N];
        gemv<R, N, N, true, 0>(J, msb, db);
        gemm<R, N, N, N, true, false, 0>(J, Psb, T1);
        gemm<R, N, N, N, false, false, 0>(T1, J, Db);
        // J = Pf W ; W = An^T Pi
        gemm<R, N, N, N, false, true, 1>(Jb, W, Pfb);      // Pfb += Jb W^T
        R Wb[N * N], Anb[N * N], Pib[N * N], Pnb[N * N];
        gemm<R, N, N, N, true, false, 0>(Pf, Jb, Wb);      // Pf^T Jb
        gemm<R, N, N, N, false, true, 0>(Pi, Wb, Anb);     // Anb = Pi Wb^T
        gemm<R, N, N, N, false, false, 0>(An, Wb, Pib);    // Pib = An Wb
        // Pnb = -Db - Pi^T Pib Pi^T
        gemm<R, N, N, N, true, false, 0>(Pi, Pib, T1);
        gemm<R, N, N, N, false, true, 0>(T1, Pi, Pnb);
        KV_UNROLL
        for (int i = 0; i < N * N; ++i) Pnb[i] = -Pnb[i] - Db[i];
        R mnb[N];
        KV_UNROLL
        for (int i = 0; i < N; ++i) mnb[i] = -db[i];
        gstore<R, N * N>(g.d_P + tb * N * N, Pfb);
        gstore<R, N>(g.d_mn + tb * N, mnb);
        gstore<R, N * N>(g.d_Pn + tb * N * N, Pnb);
        gstore<R, N * N>(g.d_As + (tb + B) * N * N, Anb);
        // carry to step t+1
        vcopy<R, N>(db, msb);
        vcopy<R, N * N>(Db, Psb);
    }
    {
        // last step: smoothed == filtered, emission uses C_T
        const long tb = (long)(T - 1) * B + b;
        R ms0[N], Ps0[N * N], C[P * N], Cb[P * N];
        gload<R, N>(f.ms + tb * N, ms0);
        gload<R, N * N>(f.Ps + tb * N * N, Ps0);
        gload<R, P * N>(f.Cs + (tb + B) * P * N, C);
        if (g.g_ms) {
            KV_UNROLL
            for (int i = 0; i < N; ++i) msb[i] += g.g_ms[tb * N + i];
        }
        if (g.g_Ps) {
            KV_UNROLL
            for (int i = 0; i < N * N; ++i) Psb[i] += g.g_Ps[tb * N * N + i];
        }
        smooth_emit_adjoint<R, N, P>(g, tb, ms0, Ps0, C, msb, Psb, Cb, acc_LR);
        gstore<R, P * N>(g.d_Cs + (tb + B) * P * N, Cb);
        gstore<R, N>(g.d_m + tb * N, msb);
        gstore<R, N * N>(g.d_P + tb * N * N, Psb);
    }
}

// ---------------------------------------------------------------------------
// Kalman ELBO terms.  One lane per (t, b) pair: the three terms of a pair need
// z_t and (for the transition term) z_{t-1}; the backward additionally needs
// z_{t+1}.  Neighbouring samples are recomputed from (ms, Ps, eps) rather than
// exchanged, so no lane depends on another.
// ---------------------------------------------------------------------------
template <typename R>
KV_HD R log_2pi() {
    return R(1.8378770664093454835606594728112);
}

// z = ms + chol(Ps) eps at pair tb; optionally returns L and eps.
template <typename R, int N>
KV_HD bool elbo_sample(const ElboFwdArgs<R>& g, long tb, R* z, R* L_out = nullptr,
                       R* eps_out = nullptr) {
    R Ps[N * N], L[N * N], e[N];
    gload<R, N * N>(g.Ps + tb * N * N, Ps);
    gload<R, N>(g.ms + tb * N, z);
    gload<R, N>(g.eps + tb * N, e);
    const bool ok = cholesky<R, N>(Ps, L);
    gemv<R, N, N, false, 1>(L, e, z);
    if (L_out) vcopy<R, N * N>(L, L_out);
    if (eps_out) vcopy<R, N>(e, eps_out);
    return ok;
}

// -1/2 (d log 2pi + |L^-1 r|^2) - sum log diag L ; r is overwritten by
// u = Sigma^-1 r (the gradient direction) when want_u.
template <typename R, int D>
KV_HD R gauss_logpdf(const R* L, R* r, bool want_u) {
    solve_lower<R, D>(L, r);
    R q = vdot<R, D>(r, r);
    R hl = R(0);
    KV_UNROLL
    for (int i = 0; i < D; ++i) hl += log_pos(L[i * D + i]);
    if (want_u) solve_lower_t<R, D>(L, r);
    return R(-0.5) * (R(D) * log_2pi<R>() + q) - hl;
}

template <typename R, int N, int P>
KV_HD void elbo_forward_pair(const ElboFwdArgs<R>& g, int t, int b) {
    const long B = g.B;
    const long tb = (long)t * B + b;
    int st = 0;
    R z[N], L[N * N], e[N];
    if (!elbo_sample<R, N>(g, tb, z, L, e)) st |= ST_NONPD_SMOOTHED;
    if (g.z) gstore<R, N>(g.z + tb * N, z);

    // observation term
    {
        R C[P * N], Rm[P * P], LR[P * P], r[P];
        gload<R, P * N>(g.Cs + tb * P * N, C);
        gload<R, P * P>(g.Rm, Rm);
        gload<R, P>(g.a + tb * P, r);
        gemv<R, P, N, false, 2>(C, z, r);
        if (!cholesky<R, P>(Rm, LR)) st |= ST_NONPD_NOISE;
        g.obs[tb] = gauss_logpdf<R, P>(LR, r, false);
    }
    // transition term
    {
        R S[N * N], LS[N * N], r[N];
        vcopy<R, N>(z, r);
        if (t == 0) {
            R m0[N];
            gload<R, N>(g.m0, m0);
            gload<R, N * N>(g.P0, S);
            KV_UNROLL
            for (int i = 0; i < N; ++i) r[i] -= m0[i];
        } else {
            R zp[N], A[N * N];
            if (!elbo_sample<R, N>(g, tb - B, zp)) st |= ST_NONPD_SMOOTHED;
            gload<R, N * N>(g.As + tb * N * N, A);
            gload<R, N * N>(g.Q, S);
            gemv<R, N, N, false, 2>(A, zp, r);
        }
        if (!cholesky<R, N>(S, LS)) st |= ST_NONPD_NOISE;
        g.trans[tb] = gauss_logpdf<R, N>(LS, r, false);
    }
    // posterior term: L^-1 (z - ms) == eps
    {
        R hl = R(0);
        KV_UNROLL
        for (int i = 0; i < N; ++i) hl += log_pos(L[i * N + i]);
        g.post[tb] = R(-0.5) * (R(N) * log_2pi<R>() + vdot<R, N>(e, e)) - hl;
    }
    if (st && g.status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(g.status, st);
#else
        *g.status |= st;
#endif
    }
}

// Gradient of w * log N(x; mu, S) w.r.t. S (symmetric form), added to Sacc:
//   w * (1/2 u u^T - 1/2 S^-1),  u = S^-1 (x - mu),  S^-1 = Li^T Li.
template <typename R, int D>
KV_HD void gauss_cov_grad_add(R w, const R* u, const R* L, R* Sacc) {
    R Li[D * D], Si[D * D];
    tri_lower_inverse<R, D>(L, Li);
    gemm<R, D, D, D, true, false, 0>(Li, Li, Si);
    KV_UNROLL
    for (int i = 0; i < D; ++i) {
        KV_UNROLL
        for (int j = 0; j < D; ++j)
            Sacc[i * D + j] += w * R(0.5) * (u[i] * u[j] - Si[i * D + j]);
    }
}

template <typename R, int N, int P>
KV_HD void elbo_backward_pair(const ElboBwdArgs<R>& g, int t, int b, R* acc) {
    using PG = ElboParamGrads<N, P>;
    const ElboFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const long tb = (long)t * B + b;
    R wo, wt, wp;
    elbo_cotangents<R>(g, tb, wo, wt, wp);

    R z[N], L[N * N], e[N];
    elbo_sample<R, N>(f, tb, z, L, e);
    R zb[N];
    vzero<R, N>(zb);

    // observation term: x = a_t, mean = C z
    {
        R C[P * N], Rm[P * P], LR[P * P], u[P];
        gload<R, P * N>(f.Cs + tb * P * N, C);
        gload<R, P * P>(f.Rm, Rm);
        gload<R, P>(f.a + tb * P, u);
        gemv<R, P, N, false, 2>(C, z, u);
        cholesky<R, P>(Rm, LR);
        gauss_logpdf<R, P>(LR, u, true);  // u = R^-1 r
        R da[P], Cb[P * N];
        KV_UNROLL
        for (int i = 0; i < P; ++i) da[i] = -wo * u[i];
        gstore<R, P>(g.d_a + tb * P, da);
        vzero<R, P * N>(Cb);
        outer_add<R, P, N>(wo, u, z, Cb);  // d mean = +w u ; mean = C z
        gstore<R, P * N>(g.d_Cs + tb * P * N, Cb);
        KV_UNROLL
        for (int j = 0; j < N; ++j) {
            R s = R(0);
            KV_UNROLL
            for (int i = 0; i < P; ++i) s += C[i * N + j] * u[i];
            zb[j] += wo * s;
        }
        gauss_cov_grad_add<R, P>(wo, u, LR, acc + PG::OFF_R);
    }
    // transition term of this pair: x = z_t
    {
        R S[N * N], LS[N * N], u[N], Ab[N * N];
        vcopy<R, N>(z, u);
        vzero<R, N * N>(Ab);
        if (t == 0) {
            R m0[N];
            gload<R, N>(f.m0, m0);
            gload<R, N * N>(f.P0, S);
            KV_UNROLL
            for (int i = 0; i < N; ++i) u[i] -= m0[i];
            cholesky<R, N>(S, LS);
            gauss_logpdf<R, N>(LS, u, true);
            KV_UNROLL
            for (int i = 0; i < N; ++i) acc[PG::OFF_M0 + i] += wt * u[i];
            gauss_cov_grad_add<R, N>(wt, u, LS, acc + PG::OFF_P0);
        } else {
            R zp[N], A[N * N];
            elbo_sample<R, N>(f, tb - B, zp);
            gload<R, N * N>(f.As + tb * N * N, A);
            gload<R, N * N>(f.Q, S);
            gemv<R, N, N, false, 2>(A, zp, u);
            cholesky<R, N>(S, LS);
            gauss_logpdf<R, N>(LS, u, true);
            outer_add<R, N, N>(wt, u, zp, Ab);
            gauss_cov_grad_add<R, N>(wt, u, LS, acc + PG::OFF_Q);
        }
        KV_UNROLL
        for (int i = 0; i < N; ++i) zb[i] -= wt * u[i];
        gstore<R, N * N>(g.d_As + tb * N * N, Ab);
    }
    // transition term of the next pair: z_t enters its mean A_{t+1} z_t
    if (t + 1 < T) {
        R wo1, wt1, wp1;   // the next pair's transition term also depends on this sample
        elbo_cotangents<R>(g, tb + B, wo1, wt1, wp1);
        (void)wo1;
        (void)wp1;
        R zn[N], A[N * N], Q[N * N], LQ[N * N], u[N];
        elbo_sample<R, N>(f, tb + B, zn);
        gload<R, N * N>(f.As + (tb + B) * N * N, A);
        gload<R, N * N>(f.Q, Q);
        vcopy<R, N>(zn, u);
        gemv<R, N, N, false, 2>(A, z, u);
        cholesky<R, N>(Q, LQ);
        gauss_logpdf<R, N>(LQ, u, true);
        KV_UNROLL
        for (int j = 0; j < N; ++j) {
            R s = R(0);
            KV_UNROLL
            for (int i = 0; i < N; ++i) s += A[i * N + j] * u[i];
            zb[j] += wt1 * s;
        }
    }
    // z = ms + L eps ; posterior term contributes -wp * sum log diag L only
    gstore<R, N>(g.d_ms + tb * N, zb);
    {
        R Lb[N * N], Psb[N * N];
        vzero<R, N * N>(Lb);
        vzero<R, N * N>(Psb);
        outer_add<R, N, N>(R(1), zb, e, Lb);
        KV_UNROLL
        for (int i = 0; i < N; ++i) Lb[i * N + i] -= wp * recip(L[i * N + i]);
        cholesky_adjoint_add<R, N>(L, Lb, Psb);
        gstore<R, N * N>(g.d_Ps + tb * N * N, Psb);
    }
    if (t == T - 1) {
        // entries of the (T+1)-long tensors that no term touches
        R z1[N * N], z2[P * N];
        vzero<R, N * N>(z1);
        vzero<R, P * N>(z2);
        gstore<R, N * N>(g.d_As + ((long)T * B + b) * N * N, z1);
        gstore<R, P * N>(g.d_Cs + ((long)T * B + b) * P * N, z2);
    }
}

}  // namespace kvae
