// "Row" form of the fused Kalman objective: ONE SEQUENCE OVER THE 16 LANES OF A DPP ROW.
//
// Lane l = 4 i + j of a row holds element (i, j) of every (at most 4 x 4) matrix of the
// sequence, so a matrix is ONE register, nothing is replicated, and a wavefront carries
// four sequences.  4096 sequences are 1024 wavefronts, one per SIMD of the chip, where
// four lanes per sequence (kalman_loss_quad.h) leave three quarters of the SIMDs without
// work and one lane per sequence (kalman_loss.h) all but 64 of them.
//
//   * products: C = X Y = sum_d X[i][(i+d)%4] Y[(i+d)%4][j]; the second factor is Y rotated
//     by 4 d lanes (row_ror), the first the quad broadcast of lane d of X "skewed"
//     (row i rotated left by i: three bank-masked quad_perm moves);
//   * X v and X^T v are one multiply and two DPP additions (over the quad / across quads);
//     vectors live either as v[j] in every quad ("c") or as v[i] in every lane of quad i ("r");
//   * the SPD inverse is the symmetric sweep operator (one pivot per step, everything
//     element-parallel), the Cholesky factor a right-looking factorisation;
//   * smaller n, p are zero / identity padded to 4: the padding block stays exactly
//     decoupled (block-diagonal operands), contributes log 1 = 0 and 0^2 to every term, and
//     is never written out.
//
// One pass per sequence runs filter -> smoother -> three log-density terms and, when GRAD,
// the complete adjoint (kalman_loss.h explains the four phases).  The per-step state the
// later phases read again lives in LDS (448 B per sequence-step): nothing but a, alpha, eps
// is read from and nothing but the term sums, d_a, d_alpha and per-wavefront parameter
// sums is written to global memory.  No covariance ever goes through HBM.
//
// The math is written once on a value type V: `float` on the device (one lane's element),
// `Row<R>` (all 16 lanes, R = float or double) in the test-only host build, where the lane
// exchanges are permutations of an array.
//
// Reference statements: kvae/state_space_model.py:284-352 (filter), :424-446 (smoother),
// kvae/kalman_vae.py:165-260 (sample, terms, weighted sum); adjoints by hand, checked against
// autograd of the oracle.
#pragma once

#include <type_traits>

#include "kalman_loss.h"

namespace kvae {
namespace row {

#if defined(__HIPCC__)
#define KR_FN __device__ __forceinline__
#else
#define KR_FN inline
#endif

template <int I, int N, typename F>
KR_FN void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- lane masks (bit l = 4 i + j) -----------------------------------------------------
constexpr unsigned M_ALL = 0xFFFFu;
constexpr unsigned M_DIAG = 0x8421u;    // i == j
constexpr unsigned M_LOWER = 0xF731u;   // i >= j
constexpr unsigned m_col(int k) { return 0x1111u << k; }
constexpr unsigned m_row(int k) { return 0xFu << (4 * k); }
constexpr unsigned m_rows_ge(int k) { return (0xFFFFu << (4 * k)) & 0xFFFFu; }
constexpr unsigned m_rows_lt(int n) { return n >= 4 ? 0xFFFFu : ((1u << (4 * n)) - 1u); }
constexpr unsigned m_diag_lt(int n) { return M_DIAG & m_rows_lt(n); }

// ---- layouts of a matrix over the 16 lanes ----------------------------------------------
enum Layout : int {
    L_NAT = 0,    // lane (i, j) holds X[i][j]
    L_TRN = 1,    // X[j][i]
    L_SKEW = 2,   // X[i][(i + j) % 4]        (left operand of a product)
    L_TSKEW = 3,  // X^T skewed: X[(i + j) % 4][i]
};
constexpr int lay_r(int lay, int i, int j) {
    return lay == L_NAT ? i : lay == L_TRN ? j : lay == L_SKEW ? i : (i + j) % 4;
}
constexpr int lay_c(int lay, int i, int j) {
    return lay == L_NAT ? j : lay == L_TRN ? i : lay == L_SKEW ? (i + j) % 4 : i;
}

// =========================================================================================
// Host value type: the 16 lanes of one sequence as an array (test-only build).
// =========================================================================================
template <typename R>
struct Row {
    R v[16];
    Row() = default;
    Row(R s) {
        for (int l = 0; l < 16; ++l) v[l] = s;
    }
};
#define KR_ROW_BINOP(op)                                                  \
    template <typename R>                                                 \
    inline Row<R> operator op(const Row<R>& a, const Row<R>& b) {         \
        Row<R> r;                                                         \
        for (int l = 0; l < 16; ++l) r.v[l] = a.v[l] op b.v[l];           \
        return r;                                                         \
    }                                                                     \
    template <typename R>                                                 \
    inline Row<R> operator op(const Row<R>& a, R b) { return a op Row<R>(b); } \
    template <typename R>                                                 \
    inline Row<R> operator op(R a, const Row<R>& b) { return Row<R>(a) op b; }
KR_ROW_BINOP(+)
KR_ROW_BINOP(-)
KR_ROW_BINOP(*)
#undef KR_ROW_BINOP
template <typename R>
inline Row<R> operator-(const Row<R>& a) {
    Row<R> r;
    for (int l = 0; l < 16; ++l) r.v[l] = -a.v[l];
    return r;
}
template <typename R>
inline Row<R>& operator+=(Row<R>& a, const Row<R>& b) { return a = a + b; }
template <typename R>
inline Row<R>& operator-=(Row<R>& a, const Row<R>& b) { return a = a - b; }

template <typename V>
struct scalar_of { using type = V; };
template <typename R>
struct scalar_of<Row<R>> { using type = R; };

// lane l of the result reads lane src(l) of x
template <typename R, typename F>
inline Row<R> permute(const Row<R>& x, F&& src) {
    Row<R> r;
    for (int l = 0; l < 16; ++l) r.v[l] = x.v[src(l)];
    return r;
}
// The lane exchanges.  These definitions ARE the specification of the device forms below.
template <int K, typename R>
inline Row<R> qb(const Row<R>& x) {            // (i, j) <- (i, K)
    return permute(x, [](int l) { return (l & 12) | K; });
}
template <int D, typename R>
inline Row<R> rot(const Row<R>& x) {           // (i, j) <- ((i + D) % 4, j)
    return permute(x, [](int l) { return (l + 4 * D) & 15; });
}
template <int L, typename R>
inline Row<R> rbc(const Row<R>& x) {           // every lane <- lane L
    return permute(x, [](int) { return L; });
}
template <typename R>
inline Row<R> skew(const Row<R>& x) {          // (i, j) <- (i, (i + j) % 4)
    return permute(x, [](int l) { return (l & 12) | (((l >> 2) + l) & 3); });
}
template <typename R>
inline Row<R> tr(const Row<R>& x) {            // (i, j) <- (j, i)
    return permute(x, [](int l) { return ((l & 3) << 2) | (l >> 2); });
}
template <typename R>
inline Row<R> qsum(const Row<R>& x) {          // sum over j, in every lane of quad i
    Row<R> r;
    for (int l = 0; l < 16; ++l) {
        const int q = l & 12;
        r.v[l] = (x.v[q] + x.v[q + 1]) + (x.v[q + 2] + x.v[q + 3]);
    }
    return r;
}
template <typename R>
inline Row<R> csum(const Row<R>& x) {          // sum over i, in every lane of column j
    Row<R> r;
    for (int l = 0; l < 16; ++l) {
        const int j = l & 3;
        r.v[l] = (x.v[j] + x.v[j + 4]) + (x.v[j + 8] + x.v[j + 12]);
    }
    return r;
}
template <unsigned MASK, typename R>
inline Row<R> sel(const Row<R>& a, const Row<R>& b) {   // lanes of MASK take a
    Row<R> r;
    for (int l = 0; l < 16; ++l) r.v[l] = ((MASK >> l) & 1u) ? a.v[l] : b.v[l];
    return r;
}
template <typename R>
inline Row<R> vrcp(const Row<R>& x) {
    Row<R> r;
    for (int l = 0; l < 16; ++l) r.v[l] = recip(x.v[l]);
    return r;
}
template <typename R>
inline Row<R> vrsq(const Row<R>& x) {
    Row<R> r;
    for (int l = 0; l < 16; ++l) r.v[l] = rsqrt_pos(x.v[l]);
    return r;
}
// log of the lanes in MASK (positive there), 0 elsewhere
template <unsigned MASK, typename R>
inline Row<R> vlog_masked(const Row<R>& x) {
    Row<R> r;
    for (int l = 0; l < 16; ++l) r.v[l] = ((MASK >> l) & 1u) ? log_pos(x.v[l]) : R(0);
    return r;
}
// a pivot: x where positive, else 1 and the status bit
template <typename R>
inline Row<R> guard_pivot(const Row<R>& x, int& st, int bit) {
    Row<R> r;
    for (int l = 0; l < 16; ++l) {
        if (x.v[l] > R(0)) {
            r.v[l] = x.v[l];
        } else {
            r.v[l] = R(1);
            st |= bit;
        }
    }
    return r;
}

// =========================================================================================
// Device value type: float, the element of this lane.
// =========================================================================================
#if defined(__HIPCC__)
template <int CTRL>
KR_FN float dpp_all(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// lanes of the quads in BANKS (bit = quad index i) take the exchanged value, the others keep `old`
template <int CTRL, int BANKS>
KR_FN float dpp_banks(float old, float x) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                          __builtin_bit_cast(int, x), CTRL, 0xf,
                                                          BANKS, false));
}
KR_FN int lane16() { return threadIdx.x & 15; }

template <int K>
KR_FN float qb(float x) { return dpp_all<K * 0x55>(x); }             // quad_perm [K,K,K,K]
template <int D>
KR_FN float rot(float x) {                                           // row_ror reads lane l - n
    if constexpr ((D & 3) == 0) return x;
    else return dpp_all<0x120 + ((16 - 4 * (D & 3)) & 15)>(x);
}
template <int L>
KR_FN float rbc(float x) { return dpp_all<0x150 + L>(x); }           // row_newbcast:L
KR_FN float skew(float x) {
    float y = x;
    y = dpp_banks<0x39, 0x2>(y, x);   // quad 1: quad_perm [1,2,3,0]
    y = dpp_banks<0x4E, 0x4>(y, x);   // quad 2: [2,3,0,1]
    y = dpp_banks<0x93, 0x8>(y, x);   // quad 3: [3,0,1,2]
    return y;
}
template <unsigned MASK>
KR_FN float sel(float a, float b) {
    if constexpr ((MASK & 0xFFFFu) == 0xFFFFu) return a;
    else if constexpr ((MASK & 0xFFFFu) == 0u) return b;
    else return ((MASK >> lane16()) & 1u) ? a : b;
}
KR_FN float tr(float x) {
    // swap the low bits of i and j (lanes l +- 3), then the high bits (l +- 6)
    const float a = sel<0x0A0Au>(dpp_all<0x12D>(x), sel<0x5050u>(dpp_all<0x123>(x), x));
    return sel<0x00CCu>(dpp_all<0x12A>(a), sel<0x3300u>(dpp_all<0x126>(a), a));
}
KR_FN float qsum(float x) {
    x += dpp_all<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_all<0x4E>(x);   // quad_perm [2,3,0,1]
    return x;
}
KR_FN float csum(float x) {
    x += dpp_all<0x124>(x);  // row_ror:4
    x += dpp_all<0x128>(x);  // row_ror:8
    return x;
}
KR_FN float vrcp(float x) { return recip(x); }
KR_FN float vrsq(float x) { return rsqrt_pos(x); }
template <unsigned MASK>
KR_FN float vlog_masked(float x) {
    return sel<MASK>(log_pos(sel<MASK>(x, 1.f)), 0.f);
}
KR_FN float guard_pivot(float x, int& st, int bit) {
    const bool ok = x > 0.f;
    st |= ok ? 0 : bit;
    return ok ? x : 1.f;
}
#endif  // __HIPCC__

// =========================================================================================
// Linear algebra on the row layout (generic in V).
// =========================================================================================
template <typename V>
KR_FN V rsum(V x) { return csum(qsum(x)); }               // sum over the 16 lanes
template <typename V>
KR_FN V r2c(V xr) { return csum(sel<M_DIAG>(xr, V(0))); }   // v[i] per quad -> v[j] per column
template <typename V>
KR_FN V c2r(V xc) { return qsum(sel<M_DIAG>(xc, V(0))); }

// X Y from X skewed and the four rotations of Y
template <typename V>
KR_FN V mm_sr(V Xs, V Y0, V Y1, V Y2, V Y3) {
    return (qb<0>(Xs) * Y0 + qb<1>(Xs) * Y1) + (qb<2>(Xs) * Y2 + qb<3>(Xs) * Y3);
}
template <typename V>
KR_FN V mm_s(V Xs, V Y) { return mm_sr(Xs, Y, rot<1>(Y), rot<2>(Y), rot<3>(Y)); }
template <typename V>
KR_FN V mm(V X, V Y) { return mm_s(skew(X), Y); }
template <typename V>
KR_FN V symm(V X) {
    using S = typename scalar_of<V>::type;
    return S(0.5) * (X + tr(X));
}

// Inverse of the leading DIM x DIM block of a symmetric positive definite X by the
// symmetric sweep operator: pivot k turns x_kk into -1/x_kk, row / column k into x_ik/x_kk
// and the rest into x_ij - x_ik x_kj / x_kk; after all pivots the block holds -X^-1.
// The remaining (padding) block must be decoupled from the leading one and keeps its
// values up to sign.  hl = sum_k 1/2 log(pivot_k) = 1/2 log det.
template <int DIM, typename V>
KR_FN V spd_inverse(V X, V& hl, int& st, int bit) {
    using S = typename scalar_of<V>::type;
    V logs = V(S(0));
    static_for<0, DIM>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const V d = guard_pivot(rbc<5 * k>(X), st, bit);
        logs += vlog_masked<M_ALL>(d);
        const V r = vrcp(d);
        const V colr = qb<k>(X);          // X[i][k]
        const V colc = r2c(colr);         // X[j][k] = X[k][j]
        const V t = colr * r;
        const V upd = X - t * colc;
        const V edge = sel<m_col(k)>(t, colc * r);
        X = sel<m_col(k) | m_row(k)>(sel<m_col(k) & m_row(k)>(-r, edge), upd);
    });
    hl = S(0.5) * logs;
    return -X;
}

// Lower Cholesky factor of the leading DIM x DIM block (identity on the padding block).
template <int DIM, typename V>
KR_FN V cholesky_row(V X, int& st, int bit) {
    using S = typename scalar_of<V>::type;
    V L = sel<M_DIAG>(V(S(1)), V(S(0)));
    static_for<0, DIM>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const V d = guard_pivot(rbc<5 * k>(X), st, bit);
        const V inv = vrsq(d);
        const V colr = sel<m_rows_ge(k)>(qb<k>(X) * inv, V(S(0)));   // L[i][k], i >= k
        const V colc = r2c(colr);
        X = X - colr * colc;
        L = sel<m_col(k)>(colr, L);
    });
    return L;
}

// (L^-1)^T for the leading DIM x DIM block of a lower-triangular L (identity elsewhere):
// columns of X = U^-1, U = L^T, left to right:  X[:,k] = (e_k - sum_{q<k} X[:,q] U[q][k]) / U[k][k].
template <int DIM, typename V>
KR_FN V tri_inverse_t(V L) {
    using S = typename scalar_of<V>::type;
    const V eye = sel<M_DIAG>(V(S(1)), V(S(0)));
    V rest = eye, X = eye;
    static_for<0, DIM>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const V inv = vrcp(rbc<5 * k>(L));
        X = sel<m_col(k)>(rest * inv, X);
        const V colc = r2c(qb<k>(L));     // L[j][k] = U[k][j]
        rest = rest - qb<k>(X) * colc;
    });
    return X;
}

// =========================================================================================
// Per-step state in LDS (float offsets inside one sequence-step; a matrix is 16 floats,
// a vector 4).  The adjoint phases overwrite slots the forward phases no longer need.
// =========================================================================================
template <bool GRAD>
struct Slots;
template <>
struct Slots<true> {
    static constexpr int V_M = 0, V_MN = 4, V_D = 8, V_Z = 12;
    static constexpr int M_PFILT = 16, M_PNEXT = 32, M_D = 48, M_L = 64, M_J = 80, M_PINV = 96;
    static constexpr int STEP = 112;
    // gradients w.r.t. the filter's outputs, written by the smoother adjoint
    static constexpr int V_BM = V_D, V_BMN = V_Z, M_BP = M_D, M_BPN = M_L;
};
template <>
struct Slots<false> {
    static constexpr int V_M = 0, V_MN = 4, M_PFILT = 8, M_PNEXT = 24;
    static constexpr int STEP = 40;
    static constexpr int V_D = 0, V_Z = 0, M_D = 0, M_L = 0, M_J = 0, M_PINV = 0;   // unused
    static constexpr int V_BM = 0, V_BMN = 0, M_BP = 0, M_BPN = 0;
};

// running sums of the parameter gradients of one sequence (device: of this lane)
template <typename V, int K>
struct RowAcc {
    V AK[K];    // d A_K[k], natural layout
    V AKT[K];   // a second part of d A_K[k], transposed layout
    V CK[K];
    V Q, Rm, P0;
    V m0c;      // d m0 as v[j]
};
template <typename V, int K>
KR_FN void zero_acc(RowAcc<V, K>& a) {
    using S = typename scalar_of<V>::type;
    const V z = V(S(0));
    for (int k = 0; k < K; ++k) a.AK[k] = a.AKT[k] = a.CK[k] = z;
    a.Q = a.Rm = a.P0 = a.m0c = z;
}

template <typename V, int K>
KR_FN V mix_row(const V* w, const V* bases) {
    V acc = w[0] * bases[0];
    KV_UNROLL
    for (int k = 1; k < K; ++k) acc = acc + w[k] * bases[k];
    return acc;
}

// =========================================================================================
// One sequence: filter -> smoother -> terms (-> adjoint).
//
// Env supplies the memory accesses (device: this lane's share; host: all 16 lanes):
//   scalar(p)                      *p in every lane
//   vec_c<LEN>(p) / vec_r<LEN>(p)  p[j] / p[i], zero beyond LEN
//   mat<ROWS, COLS, LAYOUT, PAD>(p) row-major ROWS x COLS matrix, zero (PAD: identity) padded
//   put_scalar(p, x), put_vec_c<LEN>(p, xc)              (nothing for an inactive sequence)
//   st_mat / ld_mat<LAYOUT> / st_vec / ld_vec_c / ld_vec_r (t, slot)   the LDS state
//   fence()                        orders the LDS accesses of the lanes of the sequence
// sums[0..2] += obs, trans, post of the sequence (value in every lane).
// =========================================================================================
template <typename V, typename R, int N, int P, int K, bool GRAD, typename Env>
KR_FN void loss_row_sequence(Env& env, const LossBwdArgs<R>& g, long b, V* sums,
                             RowAcc<V, K>& acc, int& st) {
    using SL = Slots<GRAD>;
    const LossFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const bool sym = f.symmetrize != 0;
    const R half = R(0.5);
    const V zero = V(R(0));

    // ---- constants of the call -------------------------------------------------------
    const V Q = env.template mat<N, N, L_NAT, true>(f.Q);
    const V Rm = env.template mat<P, P, L_NAT, true>(f.Rm);
    const V P0 = env.template mat<N, N, L_NAT, true>(f.P0);
    const V m0c = env.template vec_c<N>(f.m0), m0r = env.template vec_r<N>(f.m0);
    V hlR, hlQ, hl0;
    const V Ri = spd_inverse<P>(Rm, hlR, st, ST_NONPD_NOISE);
    const V Qi = spd_inverse<N>(Q, hlQ, st, ST_NONPD_NOISE);
    const V P0i = spd_inverse<N>(P0, hl0, st, ST_NONPD_NOISE);
    V AK[K], AKs[K], AKT[K], AKTs[K], CK[K], CKs[K], CKT[K], CKTs[K];
    KV_UNROLL
    for (int k = 0; k < K; ++k) {
        AK[k] = env.template mat<N, N, L_NAT, false>(f.A_K + k * N * N);
        AKs[k] = env.template mat<N, N, L_SKEW, false>(f.A_K + k * N * N);
        AKT[k] = env.template mat<N, N, L_TRN, false>(f.A_K + k * N * N);
        AKTs[k] = env.template mat<N, N, L_TSKEW, false>(f.A_K + k * N * N);
        CK[k] = env.template mat<P, N, L_NAT, false>(f.C_K + k * P * N);
        CKs[k] = env.template mat<P, N, L_SKEW, false>(f.C_K + k * P * N);
        CKT[k] = env.template mat<P, N, L_TRN, false>(f.C_K + k * P * N);
        if (GRAD) CKTs[k] = env.template mat<P, N, L_TSKEW, false>(f.C_K + k * P * N);
    }
    const R c_obs = R(-0.5) * R(P) * log_2pi<R>(), c_lat = R(-0.5) * R(N) * log_2pi<R>();
    auto weights = [&](long tb, V* w) {
        KV_UNROLL
        for (int k = 0; k < K; ++k) w[k] = env.scalar(f.alpha + tb * K + k);
    };

    // Quantities of one measurement update (state_space_model.py:291-311), from the
    // prediction (mp, Pp) and C in three layouts.
    struct Update {
        V M, F, Si, G, rr, rc;
    };
    auto update_terms_row = [&](V C, V Cs, V CT, V mpc, V Pp, V ar, int bit) {
        Update u;
        u.M = mm_s(Cs, Pp);                 // C Pp
        u.F = mm(Pp, CT);                   // Pp C^T
        V hl;
        u.Si = spd_inverse<P>(mm_s(Cs, u.F) + Rm, hl, st, bit);
        u.G = mm(u.F, u.Si);
        u.rr = ar - qsum(C * mpc);          // innovation a - C mp
        u.rc = r2c(u.rr);
        return u;
    };

    // Everything a step reads from global memory (observations, weights, noise, mask) is
    // independent of the recursion: every loop below requests the inputs of its next step
    // before it works on the current one, so no memory latency sits in the serial chain.
    auto clamp_t = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };

    // ---- phase F: filter, t = 0..T-1 -----------------------------------------------------
    {
        struct In {
            V wn[K], ar, mu;
        };
        auto load_in = [&](int t) {
            In x;
            const long tb = (long)t * B + b;
            weights(tb + B, x.wn);
            x.ar = env.template vec_r<P>(f.a + tb * P);
            x.mu = f.mask ? env.scalar(f.mask + tb) : V(R(1));
            return x;
        };
        V Pp = P0, mpc = m0c, mpr = m0r;
        V w[K];
        weights(b, w);
        In in = load_in(0);
        for (int t = 0; t < T; ++t) {
            const In nxt = load_in(clamp_t(t + 1));
            const V C = mix_row<V, K>(w, CK), Cs = mix_row<V, K>(w, CKs), CT = mix_row<V, K>(w, CKT);
            const V An = mix_row<V, K>(in.wn, AK), Ans = mix_row<V, K>(in.wn, AKs),
                    AnT = mix_row<V, K>(in.wn, AKT);
            const Update u = update_terms_row(C, Cs, CT, mpc, Pp, in.ar, ST_NONPD_INNOVATION);
            V Gr = qsum(u.G * u.rc), GM = mm(u.G, u.M);
            if (f.mask) {
                Gr = in.mu * Gr;
                GM = in.mu * GM;
            }
            const V mr = mpr + Gr, mc = r2c(mr);
            V Pf = Pp - GM;
            if (sym) Pf = symm(Pf);
            V Pn = mm(mm_s(Ans, Pf), AnT) + Q;          // An Pf An^T + Q
            if (sym) Pn = symm(Pn);
            mpr = qsum(An * mc);
            mpc = csum(AnT * mr);
            Pp = Pn;
            env.st_vec(t, SL::V_M, mc);
            env.st_mat(t, SL::M_PFILT, Pf);
            env.st_vec(t, SL::V_MN, mpc);
            env.st_mat(t, SL::M_PNEXT, Pn);
            KV_UNROLL
            for (int k = 0; k < K; ++k) w[k] = in.wn[k];
            in = nxt;
        }
    }
    env.fence();

    // ---- phase S: smoother and terms, t = T-1..0 -------------------------------------------
    V s_obs = zero, s_trans = zero, s_post = zero;
    R* t_obs = f.terms;
    R* t_trans = f.terms ? f.terms + (long)T * B : nullptr;
    R* t_post = f.terms ? f.terms + 2 * (long)T * B : nullptr;
    // inputs of smoother step t: weights of step t (C_t) and t+1 (A_{t+1}), noise, observation
    struct InS {
        V w[K], wn[K], ec, ar;
    };
    auto load_S = [&](int t) {
        InS x;
        const long tb = (long)t * B + b;
        weights(tb, x.w);
        weights(tb + B, x.wn);
        x.ec = env.template vec_c<N>(f.eps + tb * N);
        x.ar = env.template vec_r<P>(f.a + tb * P);
        return x;
    };
    // z_t = ms + chol(Ps) eps_t, obs_t, post_t (kalman_vae.py:165-168, 191-201, 236-238)
    auto emit = [&](int t, const InS& in, V msr, V Ps, V& zc, V& zr) {
        const long tb = (long)t * B + b;
        const V L = cholesky_row<N>(Ps, st, ST_NONPD_SMOOTHED);
        zr = msr + qsum(L * in.ec);
        zc = r2c(zr);
        const V post =
            (c_lat - half * qsum(in.ec * in.ec)) - rsum(vlog_masked<m_diag_lt(N)>(L));
        const V C = mix_row<V, K>(in.w, CK);
        const V rr = in.ar - qsum(C * zc);
        const V ur = qsum(Ri * r2c(rr));
        const V obs = (c_obs - half * csum(rr * ur)) - hlR;
        s_obs += obs;
        s_post += post;
        if (t_obs) {
            env.put_scalar(t_obs + tb, obs);
            env.put_scalar(t_post + tb, post);
        }
        if (GRAD) {
            env.st_vec(t, SL::V_Z, zc);
            env.st_mat(t, SL::M_L, L);
        }
    };
    {
        V msc = env.ld_vec_c(T - 1, SL::V_M), msr = env.ld_vec_r(T - 1, SL::V_M);
        V Ps = env.template ld_mat<L_NAT>(T - 1, SL::M_PFILT);
        V z1c, z1r;
        InS in = load_S(T - 1);
        {
            const InS nxt = load_S(clamp_t(T - 2));
            emit(T - 1, in, msr, Ps, z1c, z1r);
            in = nxt;
        }
        for (int t = T - 2; t >= 0; --t) {
            const long tb = (long)t * B + b;
            const InS nxt = load_S(clamp_t(t - 1));
            const V An = mix_row<V, K>(in.wn, AK), AnTs = mix_row<V, K>(in.wn, AKTs);
            const V Pf = env.template ld_mat<L_NAT>(t, SL::M_PFILT);
            const V Pn = env.template ld_mat<L_NAT>(t, SL::M_PNEXT);
            const V mr = env.ld_vec_r(t, SL::V_M);
            const V mnc = env.ld_vec_c(t, SL::V_MN);
            V hl;
            const V Pi = spd_inverse<N>(Pn, hl, st, ST_NONPD_PREDICTED);
            const V J = mm(Pf, mm_s(AnTs, Pi));          // Pf An^T Pn^-1
            const V dc = msc - mnc;
            const V D = Ps - Pn;
            msr = mr + qsum(J * dc);
            msc = r2c(msr);
            Ps = Pf + mm(J, tr(mm(J, D)));               // Pf + J D J^T  (D symmetric)
            if (sym) Ps = symm(Ps);
            if (GRAD) {
                env.st_vec(t, SL::V_D, dc);
                env.st_mat(t, SL::M_D, D);
                env.st_mat(t, SL::M_J, J);
                env.st_mat(t, SL::M_PINV, Pi);
            }
            V zc, zr;
            emit(t, in, msr, Ps, zc, zr);
            // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q)   (kalman_vae.py:209-229)
            const V rr = z1r - qsum(An * zc);
            const V ur = qsum(Qi * r2c(rr));
            const V tr_ = (c_lat - half * csum(rr * ur)) - hlQ;
            s_trans += tr_;
            if (t_trans) env.put_scalar(t_trans + tb + B, tr_);
            z1c = zc;
            z1r = zr;
            in = nxt;
            env.fence();
        }
        // trans_0 = log N(z_0; m0, P0)
        const V rr = z1r - m0r;
        const V ur = qsum(P0i * (z1c - m0c));
        const V tr0 = (c_lat - half * csum(rr * ur)) - hl0;
        s_trans += tr0;
        if (t_trans) env.put_scalar(t_trans + b, tr0);
    }
    sums[0] += s_obs;
    sums[1] += s_trans;
    sums[2] += s_post;
    env.fence();

    if constexpr (GRAD) {
        const R gs0 = g.g_scale ? g.g_scale[0] : R(1);
        const R gs1 = g.g_scale ? g.g_scale[g.g_scale_stride] : R(1);
        const R gs2 = g.g_scale ? g.g_scale[2 * g.g_scale_stride] : R(1);
        const R wo = g.w_obs * gs0, wt = g.w_trans * gs1, wp = g.w_post * gs2;
        const unsigned burn = g.burn_in > 0 ? (unsigned)g.burn_in : 0u;

        // ---- phase SA: adjoint of the terms and of the RTS steps, t = 0..T-1 -----------------
        {
            V zc = env.ld_vec_c(0, SL::V_Z), zr = env.ld_vec_r(0, SL::V_Z);
            // trans_0: u = P0^-1 (z_0 - m0)
            V utr = qsum(P0i * (zc - m0c)), utc = csum(P0i * (zr - m0r));
            acc.m0c += wt * utc;
            acc.P0 += (wt * half) * (utr * utc - P0i);
            V suuQ = zero, suuR = zero;
            V msbc = zero, msbr = zero, Psb = zero;
            V dA_row[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) dA_row[k] = zero;
            struct In {
                V wn[K], ec, er, ar, ac;
            };
            auto load_in = [&](int t) {
                In x;
                const long tb = (long)t * B + b;
                weights(tb + B, x.wn);
                x.ec = env.template vec_c<N>(f.eps + tb * N);
                x.er = env.template vec_r<N>(f.eps + tb * N);
                x.ar = env.template vec_r<P>(f.a + tb * P);
                x.ac = env.template vec_c<P>(f.a + tb * P);
                return x;
            };
            V w[K];
            weights(b, w);
            In in = load_in(0);
            for (int t = 0; t < T; ++t) {
                const long tb = (long)t * B + b;
                const bool last = t == T - 1;
                const In nxt = load_in(clamp_t(t + 1));
                const V* wn = in.wn;
                const V C = mix_row<V, K>(w, CK), CT = mix_row<V, K>(w, CKT);
                const V An = mix_row<V, K>(wn, AK), AnT = mix_row<V, K>(wn, AKT);
                const V L = env.template ld_mat<L_NAT>(t, SL::M_L);
                const V LT = env.template ld_mat<L_TRN>(t, SL::M_L);
                const V ec = in.ec, er = in.er;
                // obs_t = log N(a_t; C_t z_t, R):  u = R^-1 (a - C z)
                const V rr = in.ar - qsum(C * zc);
                const V rc = in.ac - csum(CT * zr);
                const V ur = qsum(Ri * rc), uc = csum(Ri * rr);
                env.template put_vec_c<P>(g.d_a + tb * P, (-wo) * uc);
                V Cb = wo * (ur * zc);
                V zbc = wo * csum(C * ur) - wt * utc;
                V zbr = wo * qsum(CT * uc) - wt * utr;
                suuR += ur * uc;
                V AbT = zero;   // gradient w.r.t. A_{t+1}, transposed
                V z1c = zero, z1r = zero, unr = zero, unc = zero;
                if (!last) {
                    // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q):  u = Q^-1 (z_{t+1} - A z_t)
                    z1c = env.ld_vec_c(t + 1, SL::V_Z);
                    z1r = env.ld_vec_r(t + 1, SL::V_Z);
                    const V qr = z1r - qsum(An * zc), qc = z1c - csum(AnT * zr);
                    unr = qsum(Qi * qc);
                    unc = csum(Qi * qr);
                    zbc += wt * csum(An * unr);
                    zbr += wt * qsum(AnT * unc);
                    AbT = wt * (zr * unc);
                    suuQ += unr * unc;
                }
                // z_t = ms_t + L eps_t, post_t = const - sum log diag L:  through the factor
                msbc += zbc;
                msbr += zbr;
                {
                    const V gc = csum(L * zbr), gr = qsum(LT * zbc);       // L^T zb
                    const V Psym = sel<M_LOWER>(gr * ec, gc * er) - sel<m_diag_lt(N)>(V(wp), zero);
                    const V LiT = tri_inverse_t<N>(L);
                    const V LiTs = skew(LiT);
                    Psb += half * mm_s(LiTs, tr(mm_s(LiTs, Psym)));         // 1/2 Li^T Psym Li
                }
                // C-part of d_alpha[t], basis gradient of C
                KV_UNROLL
                for (int k = 0; k < K; ++k) {
                    env.put_scalar(g.d_alpha + tb * K + k, dA_row[k] + rsum(Cb * CK[k]));
                    acc.CK[k] += w[k] * Cb;
                }
                if (last) {
                    env.st_vec(t, SL::V_BM, msbc);
                    env.st_mat(t, SL::M_BP, Psb);
                    env.st_vec(t, SL::V_BMN, zero);
                    env.st_mat(t, SL::M_BPN, zero);
                    break;
                }
                if ((unsigned)t < burn) {
                    msbc = msbr = zero;
                    Psb = zero;
                }
                // adjoint of the RTS step t
                const V Pf = env.template ld_mat<L_NAT>(t, SL::M_PFILT);
                const V dc = env.ld_vec_c(t, SL::V_D);
                const V D = env.template ld_mat<L_NAT>(t, SL::M_D);
                const V J = env.template ld_mat<L_NAT>(t, SL::M_J);
                const V JT = env.template ld_mat<L_TRN>(t, SL::M_J);
                const V Pi = env.template ld_mat<L_NAT>(t, SL::M_PINV);
                const V Ans = mix_row<V, K>(wn, AKs), AnTs = mix_row<V, K>(wn, AKTs);
                const V W = mm_s(AnTs, Pi);                    // An^T Pn^-1
                const V Hs = Psb + tr(Psb);
                const V H = sym ? half * Hs : Psb;
                const V Jb = msbr * dc + mm(Hs, mm(J, D));     // msb d^T + (H + H^T) J D
                const V dbc = csum(J * msbr), dbr = qsum(JT * msbc);        // J^T msb
                const V Db = mm(JT, mm(H, J));                  // J^T H J
                const V Pfb = H + mm(Jb, tr(W));
                const V Wb = mm(Pf, Jb);
                AbT += mm(Wb, Pi);                              // (Pn^-1 Wb^T)^T
                const V Pib = mm_s(Ans, Wb);
                const V Pnb = -(mm(Pi, mm(Pib, Pi))) - Db;
                env.st_vec(t, SL::V_BM, msbc);
                env.st_mat(t, SL::M_BP, Pfb);
                env.st_vec(t, SL::V_BMN, -dbc);
                env.st_mat(t, SL::M_BPN, Pnb);
                // A-part of d_alpha[t+1], basis gradient of A (transposed layout)
                KV_UNROLL
                for (int k = 0; k < K; ++k) {
                    dA_row[k] = rsum(AbT * AKT[k]);
                    acc.AKT[k] += wn[k] * AbT;
                }
                msbc = dbc;
                msbr = dbr;
                Psb = Db;
                zc = z1c;
                zr = z1r;
                utr = unr;
                utc = unc;
                KV_UNROLL
                for (int k = 0; k < K; ++k) w[k] = wn[k];
                in = nxt;
                env.fence();
            }
            // covariance gradients of the Gaussian terms: w/2 (sum u u^T - count Sigma^-1)
            acc.Rm += (wo * half) * (suuR - R(T) * Ri);
            acc.Q += (wt * half) * (suuQ - R(T - 1) * Qi);
        }
        env.fence();

        // ---- phase FA: filter adjoint, t = T-1..0 ----------------------------------------------
        {
            V mpbc = zero, mpbr = zero, Ppb = zero;
            V dwn[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) dwn[k] = zero;
            // inputs of step t: weights of step t, observation, mask, and what phase SA left
            // in d_alpha[t+1] and d_a[t]
            struct In {
                V w[K], ar, mu, da_old[K], a_old;
            };
            auto load_in = [&](int t) {
                In x;
                const long tb = (long)t * B + b;
                weights(tb, x.w);
                x.ar = env.template vec_r<P>(f.a + tb * P);
                x.mu = f.mask ? env.scalar(f.mask + tb) : V(R(1));
                KV_UNROLL
                for (int k = 0; k < K; ++k)
                    x.da_old[k] = t + 1 < T ? env.scalar(g.d_alpha + (tb + B) * K + k) : zero;
                x.a_old = env.template vec_c<P>(g.d_a + tb * P);
                return x;
            };
            V wn[K];
            weights((long)T * B + b, wn);
            In in = load_in(T - 1);
            for (int t = T - 1; t >= 0; --t) {
                const long tb = (long)t * B + b;
                const In nxt = load_in(clamp_t(t - 1));
                const V* w = in.w;
                const V ar = in.ar;
                const V C = mix_row<V, K>(w, CK), Cs = mix_row<V, K>(w, CKs),
                        CT = mix_row<V, K>(w, CKT), CTs = mix_row<V, K>(w, CKTs);
                const V An = mix_row<V, K>(wn, AK), Ans = mix_row<V, K>(wn, AKs),
                        AnT = mix_row<V, K>(wn, AKT), AnTs = mix_row<V, K>(wn, AKTs);
                V mpc, Pp;
                if (t > 0) {
                    mpc = env.ld_vec_c(t - 1, SL::V_MN);
                    Pp = env.template ld_mat<L_NAT>(t - 1, SL::M_PNEXT);
                } else {
                    mpc = m0c;
                    Pp = P0;
                }
                const V mc = env.ld_vec_c(t, SL::V_M);
                const V Pf = env.template ld_mat<L_NAT>(t, SL::M_PFILT);
                // 1. gradients of the step's four outputs
                V mbc = env.ld_vec_c(t, SL::V_BM), mbr = env.ld_vec_r(t, SL::V_BM);
                V Pb = env.template ld_mat<L_NAT>(t, SL::M_BP);
                mpbc += env.ld_vec_c(t, SL::V_BMN);
                mpbr += env.ld_vec_r(t, SL::V_BMN);
                Ppb += env.template ld_mat<L_NAT>(t, SL::M_BPN);
                if ((unsigned)t < burn) {
                    mbc = mbr = mpbc = mpbr = zero;
                    Pb = Ppb = zero;
                }
                // 2. time update  m' = An m,  P' = sym(An Pf An^T + Q)
                {
                    const V Hs = Ppb + tr(Ppb);
                    const V H = sym ? half * Hs : Ppb;
                    const V Ab = mpbr * mc + mm(Hs, mm_s(Ans, Pf));   // mpb m^T + (H + H^T) An Pf
                    mbc += csum(An * mpbr);                           // An^T mpb
                    mbr += qsum(AnT * mpbc);
                    Pb += mm_s(AnTs, mm(H, An));                      // An^T H An
                    acc.Q += H;
                    // d_alpha[t+1] is complete here
                    KV_UNROLL
                    for (int k = 0; k < K; ++k) {
                        const V da = (dwn[k] + rsum(Ab * AK[k])) + in.da_old[k];
                        env.put_scalar(g.d_alpha + (tb + B) * K + k, da);
                        acc.AK[k] += wn[k] * Ab;
                    }
                }
                // 3. measurement update
                const Update u = update_terms_row(C, Cs, CT, mpc, Pp, ar, 0);
                const V H2s = Pb + tr(Pb);
                const V H2 = sym ? half * H2s : Pb;
                V mobc = mbc, mobr = mbr, Pob = H2;
                if (f.mask) {
                    mobc = in.mu * mbc;
                    mobr = in.mu * mbr;
                    Pob = in.mu * H2;
                }
                mpbc = mbc;          // blend + identity term of m_obs
                mpbr = mbr;
                Ppb = H2;
                const V GT = tr(u.G);
                const V Gb = mobr * u.rc - mm(Pob, u.F);              // mob r^T - Pob M^T
                const V rbc_ = csum(u.G * mobr), rbr = qsum(GT * mobc);   // G^T mob
                env.template put_vec_c<P>(g.d_a + tb * P, in.a_old + rbc_);
                V Cb = -(rbr * mpc);
                mpbc -= csum(C * rbr);                                // C^T rb
                mpbr -= qsum(CT * rbc_);
                const V Mb = -mm(GT, Pob);                            // -G^T Pob
                Cb += mm(Mb, Pp);
                Ppb += mm_s(CTs, Mb);                                 // C^T Mb
                const V Fb = mm(Gb, u.Si);
                const V Sib = mm(u.M, Gb);                            // F^T Gb
                const V Sb = -mm(u.Si, mm(Sib, u.Si));
                Ppb += mm(Fb, C);
                Cb += mm(tr(Fb), Pp);
                Cb += mm(Sb + tr(Sb), u.M);                           // Sb C Pp^T + Sb^T C Pp
                Ppb += mm_s(CTs, mm(Sb, C));                          // C^T Sb C
                acc.Rm += Sb;
                // 4. C-part of d_alpha[t]
                KV_UNROLL
                for (int k = 0; k < K; ++k) {
                    dwn[k] = rsum(Cb * CK[k]);
                    acc.CK[k] += w[k] * Cb;
                }
                KV_UNROLL
                for (int k = 0; k < K; ++k) wn[k] = w[k];
                in = nxt;
            }
            // d_alpha[0]: C-part of step 0 added to what phase SA wrote
            KV_UNROLL
            for (int k = 0; k < K; ++k)
                env.put_scalar(g.d_alpha + b * K + k, dwn[k] + env.scalar(g.d_alpha + b * K + k));
            acc.m0c += mpbc;
            acc.P0 += Ppb;
        }
    }
}

}  // namespace row
}  // namespace kvae
