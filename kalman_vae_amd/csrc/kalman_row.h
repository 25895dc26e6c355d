// "Row" form of the fused Kalman objective: ONE SEQUENCE OVER THE 16 LANES OF A DPP ROW.
//
// Lane l = 4 i + j of a row holds element (i, j) of every (at most 4 x 4) matrix of the
// sequence, so a matrix is ONE register, nothing is replicated, and a wavefront carries
// four sequences.  4096 sequences are 1024 wavefronts, one per SIMD of the chip, where
// four lanes per sequence (kalman_loss_quad.h) leave three quarters of the SIMDs without
// work and one lane per sequence (kalman_loss.h) all but 64 of them.
//
//   * products: C[i][j] = sum_k X[i][k] y_k[j] is four multiply-adds whose first factor is
//     the quad broadcast of lane k of X (a DPP operand of the multiply-add itself) and whose
//     second factor y_k is row k of the right operand as seen from column j.  For an operand
//     that lives in LDS (every matrix a later phase reads again does) all four y_k are ONE
//     16-byte read of a row or column: no rotation, no transposition.  For an operand that
//     exists only in registers they are Y rotated by 4 k lanes (row_ror) against X "skewed"
//     (row i rotated left by i: three bank-masked quad_perm moves);
//   * X v and X^T v are one multiply and two DPP additions (over the quad / across quads);
//     vectors live either as v[j] in every quad ("c") or as v[i] in every lane of quad i ("r");
//   * everything of the observation's size p <= 2 (C_t, the gain, the innovation covariance
//     and their adjoints) is kept as p vectors and p x p replicated scalars, so their
//     "products" are matrix-vector and outer products and the p x p inverse is closed form;
//   * the n x n SPD inverse is the symmetric sweep operator (one pivot per step, everything
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

// Timing experiments only (tools/exp/build_row_variants.sh): a build that returns after
// phase F (1), S (2) or SA (3) of a sequence; the shipped library runs all four.
#ifndef KR_STOP_AFTER
#define KR_STOP_AFTER 4
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

// four values per lane: the rows (or columns) of a right operand as one column sees them
template <typename V>
struct Q4 {
    V v[4];
};
// C(i, j) = sum_k X(i, k) y.v[k](i, j)
template <typename R>
inline Row<R> mm4(const Row<R>& X, const Q4<Row<R>>& y) {
    return (qb<0>(X) * y.v[0] + qb<1>(X) * y.v[1]) + (qb<2>(X) * y.v[2] + qb<3>(X) * y.v[3]);
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
// The product core.  The compiler does not fold a DPP exchange into a multiply-add (it forms
// v_fmac_f32 only after its DPP pass has run) and spends a move per term, so three of the four
// instructions are written out.  A DPP operand must have been written at least two wait
// states earlier.  The compiler keeps that distance for instructions it can see: it sees the
// first term (a multiply, into which it does fold the exchange), so X is old enough when the
// asm reads it the same way, and it treats the asm's result like any other definition when a
// later DPP instruction reads it.  No fixed padding is needed (with an s_nop on either side
// of a four-instruction asm block the kernel took 73.2 instead of 72.0 us).
KR_FN float mm4(float X, const Q4<float>& y) {
    // the first term is the compiler's (it folds the exchange into a multiply and keeps X's
    // two wait states itself); the asm's later reads of X are then late enough by themselves
    float acc = qb<0>(X) * y.v[0];
    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %1, %3 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %1, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(acc)
        : "v"(X), "v"(y.v[1]), "v"(y.v[2]), "v"(y.v[3]));
    return acc;
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

// the rows of a register operand as the columns see them: Y rotated by 0, 4, 8, 12 lanes.
// mm4(skew(X), rots(Y)) = X Y.
template <typename V>
KR_FN Q4<V> rots(V Y) {
    Q4<V> q;
    q.v[0] = Y;
    q.v[1] = rot<1>(Y);
    q.v[2] = rot<2>(Y);
    q.v[3] = rot<3>(Y);
    return q;
}
template <typename V>
KR_FN V mm(V X, V Y) { return mm4(skew(X), rots(Y)); }
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
// colc(k) supplies L[j][k] in lane (i, j) (row k of U as the columns see it).
template <int DIM, typename V, typename ColC>
KR_FN V tri_inverse_t(V L, ColC&& colc) {
    using S = typename scalar_of<V>::type;
    const V eye = sel<M_DIAG>(V(S(1)), V(S(0)));
    V rest = eye, X = eye;
    static_for<0, DIM>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const V inv = vrcp(rbc<5 * k>(L));
        X = sel<m_col(k)>(rest * inv, X);
        if constexpr (k + 1 < DIM) rest = rest - qb<k>(X) * colc(kc);
    });
    return X;
}

// Inverse and half log-determinant of a symmetric positive definite P x P matrix of
// replicated scalars, P <= 2, in closed form.  S and Si are indexed [k][l]; only k <= l of S
// is read.
template <int P, typename V>
KR_FN void small_spd_inverse(const V (*S)[P], V (*Si)[P], V& hl, int& st, int bit) {
    using Sc = typename scalar_of<V>::type;
    static_assert(P == 1 || P == 2, "observation dimension of the 16-lane form is 1 or 2");
    if constexpr (P == 1) {
        const V d = guard_pivot(S[0][0], st, bit);
        Si[0][0] = vrcp(d);
        hl = Sc(0.5) * vlog_masked<M_ALL>(d);
    } else {
        const V d0 = guard_pivot(S[0][0], st, bit);
        const V det = guard_pivot(d0 * S[1][1] - S[0][1] * S[0][1], st, bit);
        const V r = vrcp(det);
        Si[0][0] = S[1][1] * r;
        Si[0][1] = Si[1][0] = -(S[0][1] * r);
        Si[1][1] = d0 * r;
        hl = Sc(0.5) * vlog_masked<M_ALL>(det);
    }
}

// =========================================================================================
// Per-step state in LDS (float offsets inside one sequence-step; a matrix is 16 floats,
// a vector 4).  The adjoint phases overwrite slots the forward phases no longer need.
// Behind the T steps there are SCRATCH matrix slots per sequence that belong to no step:
// operands that exist only in registers are bounced through them to be read as rows.
// =========================================================================================
constexpr int SCRATCH = 5;
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
template <typename V, int P, int K>
struct RowAcc {
    V AK[K];       // d A_K[k], natural layout
    V AKT[K];      // a second part of d A_K[k], transposed layout
    V CKc[K][P];   // row r of d C_K[k] as v[j]
    V Q, P0;
    V m0c;         // d m0 as v[j]
    V Rs[P][P];    // d R, replicated scalars
};
template <typename V, int P, int K>
KR_FN void zero_acc(RowAcc<V, P, K>& a) {
    using S = typename scalar_of<V>::type;
    const V z = V(S(0));
    for (int k = 0; k < K; ++k) {
        a.AK[k] = a.AKT[k] = z;
        for (int r = 0; r < P; ++r) a.CKc[k][r] = z;
    }
    a.Q = a.P0 = a.m0c = z;
    for (int r = 0; r < P; ++r)
        for (int c = 0; c < P; ++c) a.Rs[r][c] = z;
}

template <typename V, int K>
KR_FN V mix_row(const V* w, const V* bases) {
    V acc = w[0] * bases[0];
    KV_UNROLL
    for (int k = 1; k < K; ++k) acc = acc + w[k] * bases[k];
    return acc;
}
// row r of sum_k w[k] C_K[k] from bases[k][r]
template <typename V, int P, int K>
KR_FN void mix_rows(const V* w, const V (*bases)[P], V* out) {
    KV_UNROLL
    for (int r = 0; r < P; ++r) {
        V acc = w[0] * bases[0][r];
        KV_UNROLL
        for (int k = 1; k < K; ++k) acc = acc + w[k] * bases[k][r];
        out[r] = acc;
    }
}

// =========================================================================================
// One sequence: filter -> smoother -> terms (-> adjoint).
//
// Env supplies the memory accesses (device: this lane's share; host: all 16 lanes):
//   scalar(p)                      *p in every lane
//   vec_c<LEN>(p) / vec_r<LEN>(p)  p[j] / p[i], zero beyond LEN
//   mat<ROWS, COLS, LAYOUT, PAD>(p) row-major ROWS x COLS matrix, zero (PAD: identity) padded
//   put_scalar(p, x)               (nothing for an inactive sequence)
//   at_mat(t, slot) / at_vec(t, slot) / scr(i)   a place in the LDS state / scratch
//   st_mat, st_mat_t (transposed), ld_mat, ld_mat_t, ld_rows4 (lane (i, j): row j),
//   ld_cols4 (column j), ld_colc<k> (element [j][k]), st_vec, ld_vec_c, ld_vec_r
//   fence()                        orders the LDS accesses of the lanes of the sequence
// sums[0..2] += obs, trans, post of the sequence (value in every lane).
// =========================================================================================
template <typename V, typename R, int N, int P, int K, bool GRAD, bool SYM, typename Env>
KR_FN void loss_row_sequence(Env& env, const LossBwdArgs<R>& g, long b, V* sums,
                             RowAcc<V, P, K>& acc, int& st) {
    using SL = Slots<GRAD>;
    const LossFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    constexpr bool sym = SYM;   // symmetrize_covariance: compiled in, the loop bodies stay straight-line
    const R half = R(0.5);
    const V zero = V(R(0));

    // ---- constants of the call -------------------------------------------------------
    const V Q = env.template mat<N, N, L_NAT, true>(f.Q);
    const V P0 = env.template mat<N, N, L_NAT, true>(f.P0);
    const V m0c = env.template vec_c<N>(f.m0), m0r = env.template vec_r<N>(f.m0);
    V hlR, hlQ, hl0;
    const V Qi = spd_inverse<N>(Q, hlQ, st, ST_NONPD_NOISE);
    const V P0i = spd_inverse<N>(P0, hl0, st, ST_NONPD_NOISE);
    V Rs[P][P], Ri[P][P];
    KV_UNROLL
    for (int r = 0; r < P; ++r) {
        KV_UNROLL
        for (int c = 0; c < P; ++c) Rs[r][c] = env.scalar(f.Rm + r * P + c);
    }
    small_spd_inverse<P, V>(Rs, Ri, hlR, st, ST_NONPD_NOISE);
    V AK[K], AKs[K], AKT[K], AKTs[K], CKc[K][P], CKr[K][P];
    KV_UNROLL
    for (int k = 0; k < K; ++k) {
        AK[k] = env.template mat<N, N, L_NAT, false>(f.A_K + k * N * N);
        AKs[k] = env.template mat<N, N, L_SKEW, false>(f.A_K + k * N * N);
        AKT[k] = env.template mat<N, N, L_TRN, false>(f.A_K + k * N * N);
        AKTs[k] = env.template mat<N, N, L_TSKEW, false>(f.A_K + k * N * N);
        KV_UNROLL
        for (int r = 0; r < P; ++r) {
            CKc[k][r] = env.template vec_c<N>(f.C_K + (k * P + r) * N);
            CKr[k][r] = env.template vec_r<N>(f.C_K + (k * P + r) * N);
        }
    }
    const R c_obs = R(-0.5) * R(P) * log_2pi<R>(), c_lat = R(-0.5) * R(N) * log_2pi<R>();
    auto weights = [&](long tb, V* w) {
        KV_UNROLL
        for (int k = 0; k < K; ++k) w[k] = env.scalar(f.alpha + tb * K + k);
    };
    auto observation = [&](long tb, V* a) {
        KV_UNROLL
        for (int r = 0; r < P; ++r) a[r] = env.scalar(f.a + tb * P + r);
    };
    // sum_r x[r] y[r]
    auto dotp = [&](const V* x, const V* y) {
        V s = x[0] * y[0];
        KV_UNROLL
        for (int r = 1; r < P; ++r) s = s + x[r] * y[r];
        return s;
    };

    // Quantities of one measurement update (state_space_model.py:291-311) from the
    // prediction (mp, Pp) and the rows of C as v[j] (cc) and v[i] (cr):
    //   fc[r] = row r of M = C Pp,  fr[r] = column r of F = Pp C^T (as v[j] / v[i]),
    //   Si = (C Pp C^T + R)^-1,  gr / gc[r] = column r of the gain F Si,  r = a - C mp.
    struct Update {
        V fr[P], fc[P], Si[P][P], gr[P], gc[P], r[P];
    };
    auto update_terms_row = [&](const V* cc, const V* cr, V mpc, V Pp, const V* a, int bit) {
        Update u;
        KV_UNROLL
        for (int r = 0; r < P; ++r) {
            u.fr[r] = qsum(Pp * cc[r]);
            u.fc[r] = csum(Pp * cr[r]);
        }
        V S[P][P];
        KV_UNROLL
        for (int r = 0; r < P; ++r) {
            KV_UNROLL
            for (int c = r; c < P; ++c) S[r][c] = qsum(u.fc[r] * cc[c]) + Rs[r][c];
        }
        V hl;
        small_spd_inverse<P, V>(S, u.Si, hl, st, bit);
        KV_UNROLL
        for (int r = 0; r < P; ++r) {
            u.gr[r] = u.fr[0] * u.Si[0][r];
            u.gc[r] = u.fc[0] * u.Si[0][r];
            KV_UNROLL
            for (int c = 1; c < P; ++c) {
                u.gr[r] = u.gr[r] + u.fr[c] * u.Si[c][r];
                u.gc[r] = u.gc[r] + u.fc[c] * u.Si[c][r];
            }
            u.r[r] = a[r] - qsum(cc[r] * mpc);
        }
        return u;
    };

    // Everything a step reads from global memory (observations, weights, noise, mask) is
    // independent of the recursion: every loop below requests the inputs of its next step
    // before it works on the current one, so no memory latency sits in the serial chain.
    auto clamp_t = [&](int t) { return t < 0 ? 0 : (t > T - 1 ? T - 1 : t); };

    // ---- phase F: filter, t = 0..T-1 -----------------------------------------------------
    {
        struct In {
            V wn[K], a[P], mu;
        };
        auto load_in = [&](int t) {
            In x;
            const long tb = (long)t * B + b;
            weights(tb + B, x.wn);
            observation(tb, x.a);
            x.mu = f.mask ? env.scalar(f.mask + tb) : V(R(1));
            return x;
        };
        V Pp = P0, mpc = m0c, mpr = m0r;
        V w[K];
        weights(b, w);
        In in = load_in(0);
        for (int t = 0; t < T; ++t) {
            const In nxt = load_in(clamp_t(t + 1));
            V cc[P], cr[P];
            mix_rows<V, P, K>(w, CKc, cc);
            mix_rows<V, P, K>(w, CKr, cr);
            const V An = mix_row<V, K>(in.wn, AK), Ans = mix_row<V, K>(in.wn, AKs),
                    AnT = mix_row<V, K>(in.wn, AKT);
            env.st_mat(env.scr(0), An);               // read back as rows: X An^T
            const Update u = update_terms_row(cc, cr, mpc, Pp, in.a, ST_NONPD_INNOVATION);
            V Grr = u.gr[0] * u.r[0], Grc = u.gc[0] * u.r[0];
            // G M = F Si F^T is symmetric: its symmetric part directly, without a transposition
            V GM = sym ? u.gr[0] * u.fc[0] + u.gc[0] * u.fr[0] : u.gr[0] * u.fc[0];
            KV_UNROLL
            for (int r = 1; r < P; ++r) {
                Grr = Grr + u.gr[r] * u.r[r];
                Grc = Grc + u.gc[r] * u.r[r];
                GM = GM + (sym ? u.gr[r] * u.fc[r] + u.gc[r] * u.fr[r] : u.gr[r] * u.fc[r]);
            }
            const V mus = sym ? half * in.mu : in.mu;       // mask weight (1 without a mask)
            const V mr = mpr + in.mu * Grr, mc = mpc + in.mu * Grc;
            const V Pf = Pp - mus * GM;
            V Pn = mm4(mm4(Ans, rots(Pf)), env.ld_rows4(env.scr(0))) + Q;   // An Pf An^T + Q
            if (sym) Pn = symm(Pn);
            mpr = qsum(An * mc);
            mpc = csum(AnT * mr);
            Pp = Pn;
            env.st_vec(env.at_vec(t, SL::V_M), mc);
            env.st_mat(env.at_mat(t, SL::M_PFILT), Pf);
            env.st_vec(env.at_vec(t, SL::V_MN), mpc);
            env.st_mat(env.at_mat(t, SL::M_PNEXT), Pn);
            KV_UNROLL
            for (int k = 0; k < K; ++k) w[k] = in.wn[k];
            in = nxt;
            env.fence();
        }
    }

    if (KR_STOP_AFTER < 2) return;
    // ---- phase S: smoother and terms, t = T-1..0 -------------------------------------------
    V s_obs = zero, s_trans = zero, s_post = zero;
    R* t_obs = f.terms;
    R* t_trans = f.terms ? f.terms + (long)T * B : nullptr;
    R* t_post = f.terms ? f.terms + 2 * (long)T * B : nullptr;
    // inputs of smoother step t: weights of step t (C_t) and t+1 (A_{t+1}), noise, observation
    struct InS {
        V w[K], wn[K], ec, a[P];
    };
    auto load_S = [&](int t) {
        InS x;
        const long tb = (long)t * B + b;
        weights(tb, x.w);
        weights(tb + B, x.wn);
        x.ec = env.template vec_c<N>(f.eps + tb * N);
        observation(tb, x.a);
        return x;
    };
    // log N(x; mean, Sigma) from the residual in both forms and Sigma^-1: the quadratic form is
    // the sum of rr[i] Si[i][j] rc[j] over the 16 lanes
    auto gauss = [&](V rr, V rc, V Si, V hl) { return (c_lat - half * rsum(Si * rr * rc)) - hl; };
    // z_t = ms + chol(Ps) eps_t, obs_t, post_t (kalman_vae.py:165-168, 191-201, 236-238)
    auto emit = [&](int t, const InS& in, V msr, V Ps, V& zc, V& zr) {
        const long tb = (long)t * B + b;
        const V L = cholesky_row<N>(Ps, st, ST_NONPD_SMOOTHED);
        zr = msr + qsum(L * in.ec);
        zc = r2c(zr);
        const V post =
            (c_lat - half * qsum(in.ec * in.ec)) - rsum(vlog_masked<m_diag_lt(N)>(L));
        V cc[P], res[P], u[P];
        mix_rows<V, P, K>(in.w, CKc, cc);
        KV_UNROLL
        for (int r = 0; r < P; ++r) res[r] = in.a[r] - qsum(cc[r] * zc);
        KV_UNROLL
        for (int r = 0; r < P; ++r) u[r] = dotp(Ri[r], res);
        const V obs = (c_obs - half * dotp(res, u)) - hlR;
        s_obs += obs;
        s_post += post;
        if (t_obs) {
            env.put_scalar(t_obs + tb, obs);
            env.put_scalar(t_post + tb, post);
        }
        if (GRAD) {
            env.st_vec(env.at_vec(t, SL::V_Z), zc);
            env.st_mat(env.at_mat(t, SL::M_L), L);
        }
    };
    {
        V msc = env.ld_vec_c(env.at_vec(T - 1, SL::V_M)), msr = env.ld_vec_r(env.at_vec(T - 1, SL::V_M));
        V Ps = env.ld_mat(env.at_mat(T - 1, SL::M_PFILT));
        V z1c, z1r;
        InS in = load_S(T - 1);
        {
            const InS nxt = load_S(clamp_t(T - 2));
            emit(T - 1, in, msr, Ps, z1c, z1r);
            in = nxt;
        }
        // Of an RTS step only D = Ps - Pn, (J D) J^T and the mean go through the recursion;
        // Pn^-1 and the gain J = Pf An^T Pn^-1 of step t depend on the filter's results alone.
        // They are formed one iteration ahead (`pre`), so that their long dependent chain (the
        // inverse) interleaves with the Cholesky factor and the terms of the step before,
        // and the gain's rows come back from LDS long before they are multiplied.
        struct Pre {
            V Pf, Pn, mr, mnc, J;
        };
        // wn: the weights of step t + 1 (they mix A_{t+1})
        auto pre = [&](int t, const V* wn) {
            Pre p;
            p.Pf = env.ld_mat(env.at_mat(t, SL::M_PFILT));
            p.Pn = env.ld_mat(env.at_mat(t, SL::M_PNEXT));
            p.mr = env.ld_vec_r(env.at_vec(t, SL::V_M));
            p.mnc = env.ld_vec_c(env.at_vec(t, SL::V_MN));
            V hl;
            const V Pi = spd_inverse<N>(p.Pn, hl, st, ST_NONPD_PREDICTED);
            // what the adjoint reads again stays with its step, else in a scratch slot (two
            // in turn: the rows of step t's gain are read while step t - 1's is written)
            if (GRAD) env.st_mat(env.at_mat(t, SL::M_PINV), Pi);
            // (operands that have just been computed go through lane rotations, not LDS)
            const V W = mm4(mix_row<V, K>(wn, AKTs), rots(Pi));   // An^T Pn^-1
            p.J = mm4(skew(p.Pf), rots(W));                        // Pf W
            env.st_mat(GRAD ? env.at_mat(t, SL::M_J) : env.scr(1 + (t & 1)), p.J);
            return p;
        };
        auto rts_step = [&](int t, const InS& in, const Pre& p) {
            const long tb = (long)t * B + b;
            const auto Jrows = env.ld_rows4(GRAD ? env.at_mat(t, SL::M_J) : env.scr(1 + (t & 1)));
            const V An = mix_row<V, K>(in.wn, AK), AnT = mix_row<V, K>(in.wn, AKT);
            const V dc = msc - p.mnc;
            const V D = Ps - p.Pn;
            if (GRAD) {
                env.st_mat(env.at_mat(t, SL::M_D), D);
                env.st_vec(env.at_vec(t, SL::V_D), dc);
            }
            msr = p.mr + qsum(p.J * dc);
            msc = r2c(msr);
            const V JD = mm4(skew(p.J), rots(D));             // J D
            Ps = p.Pf + mm4(JD, Jrows);                       // + (J D) J^T
            if (sym) Ps = symm(Ps);
            V zc, zr;
            emit(t, in, msr, Ps, zc, zr);
            // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q)   (kalman_vae.py:209-229)
            const V tr_ = gauss(z1r - qsum(An * zc), z1c - csum(AnT * zr), Qi, hlQ);
            s_trans += tr_;
            if (t_trans) env.put_scalar(t_trans + tb + B, tr_);
            z1c = zc;
            z1r = zr;
        };
        if (T >= 2) {
            Pre cur = pre(T - 2, in.wn);
            env.fence();
            for (int t = T - 2; t >= 1; --t) {
                const InS nxt = load_S(clamp_t(t - 1));
                const Pre ahead = pre(t - 1, in.w);           // step t's weights mix A_t
                rts_step(t, in, cur);
                cur = ahead;
                in = nxt;
                env.fence();
            }
            rts_step(0, in, cur);
            env.fence();
        }
        // trans_0 = log N(z_0; m0, P0)
        const V tr0 = gauss(z1r - m0r, z1c - m0c, P0i, hl0);
        s_trans += tr0;
        if (t_trans) env.put_scalar(t_trans + b, tr0);
    }
    sums[0] += s_obs;
    sums[1] += s_trans;
    sums[2] += s_post;
    env.fence();

    if constexpr (GRAD && KR_STOP_AFTER > 2) {
        const R gs0 = g.g_scale ? g.g_scale[0] : R(1);
        const R gs1 = g.g_scale ? g.g_scale[g.g_scale_stride] : R(1);
        const R gs2 = g.g_scale ? g.g_scale[2 * g.g_scale_stride] : R(1);
        const R wo = g.w_obs * gs0, wt = g.w_trans * gs1, wp = g.w_post * gs2;
        const unsigned burn = g.burn_in > 0 ? (unsigned)g.burn_in : 0u;

        // ---- phase SA: adjoint of the terms and of the RTS steps, t = 0..T-1 -----------------
        {
            V zc = env.ld_vec_c(env.at_vec(0, SL::V_Z)), zr = env.ld_vec_r(env.at_vec(0, SL::V_Z));
            // trans_0: u = P0^-1 (z_0 - m0)
            V utr = qsum(P0i * (zc - m0c)), utc = csum(P0i * (zr - m0r));
            acc.m0c += wt * utc;
            acc.P0 += (wt * half) * (utr * utc - P0i);
            V suuQ = zero, suuR[P][P];
            KV_UNROLL
            for (int r = 0; r < P; ++r) {
                KV_UNROLL
                for (int c = 0; c < P; ++c) suuR[r][c] = zero;
            }
            V msbc = zero, msbr = zero, Psb = zero;
            V dA_row[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) dA_row[k] = zero;
            struct In {
                V wn[K], ec, er, a[P];
            };
            auto load_in = [&](int t) {
                In x;
                const long tb = (long)t * B + b;
                weights(tb + B, x.wn);
                x.ec = env.template vec_c<N>(f.eps + tb * N);
                x.er = env.template vec_r<N>(f.eps + tb * N);
                observation(tb, x.a);
                return x;
            };
            V w[K];
            weights(b, w);
            In in = load_in(0);
            // The adjoint of the terms of step t (obs_t, trans_{t+1} unless LAST, post_t, the
            // sample z_t): adds to msb / Psb, writes d_a[t] and d_alpha[t], leaves A_{t+1}'s
            // gradient from trans_{t+1} in AbT and u of trans_{t+1} in unr / unc.
            auto terms_adjoint = [&](int t, auto last_c, V& AbT, V& z1c, V& z1r, V& unr, V& unc) {
                constexpr bool LAST = decltype(last_c)::value;
                const long tb = (long)t * B + b;
                [[maybe_unused]] const V* wn = in.wn;
                V cc[P], cr[P];
                mix_rows<V, P, K>(w, CKc, cc);
                mix_rows<V, P, K>(w, CKr, cr);
                const auto locL = env.at_mat(t, SL::M_L);
                const V L = env.ld_mat(locL), LT = env.ld_mat_t(locL);
                // obs_t = log N(a_t; C_t z_t, R):  u = R^-1 (a - C z)
                V res[P], u[P], cb[P];
                KV_UNROLL
                for (int r = 0; r < P; ++r) res[r] = in.a[r] - qsum(cc[r] * zc);
                KV_UNROLL
                for (int r = 0; r < P; ++r) {
                    u[r] = dotp(Ri[r], res);
                    env.put_scalar(g.d_a + tb * P + r, (-wo) * u[r]);
                    cb[r] = (wo * u[r]) * zc;                    // row r of the gradient w.r.t. C_t
                }
                V zbc = (-wt) * utc, zbr = (-wt) * utr;
                KV_UNROLL
                for (int r = 0; r < P; ++r) {
                    zbc = zbc + (wo * u[r]) * cc[r];             // C^T u
                    zbr = zbr + (wo * u[r]) * cr[r];
                    KV_UNROLL
                    for (int c = 0; c < P; ++c) suuR[r][c] = suuR[r][c] + u[r] * u[c];
                }
                AbT = zero;   // gradient w.r.t. A_{t+1}, transposed
                if constexpr (!LAST) {
                    // trans_{t+1} = log N(z_{t+1}; A_{t+1} z_t, Q):  u = Q^-1 (z_{t+1} - A z_t)
                    const V An = mix_row<V, K>(wn, AK), AnT = mix_row<V, K>(wn, AKT);
                    z1c = env.ld_vec_c(env.at_vec(t + 1, SL::V_Z));
                    z1r = env.ld_vec_r(env.at_vec(t + 1, SL::V_Z));
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
                    const V Psym =
                        sel<M_LOWER>(gr * in.ec, gc * in.er) - sel<m_diag_lt(N)>(V(wp), zero);
                    const V LiT = tri_inverse_t<N>(L, [&](auto kc) {
                        return env.template ld_colc<decltype(kc)::value>(locL);
                    });
                    env.st_mat(env.scr(0), mm4(skew(LiT), rots(Psym)));   // V = LiT Psym
                    Psb += half * mm4(LiT, env.ld_rows4(env.scr(0)));     // 1/2 LiT V^T
                }
                // C-part of d_alpha[t], basis gradient of C
                KV_UNROLL
                for (int k = 0; k < K; ++k) {
                    env.put_scalar(g.d_alpha + tb * K + k, dA_row[k] + qsum(dotp(cb, CKc[k])));
                    KV_UNROLL
                    for (int r = 0; r < P; ++r) acc.CKc[k][r] += w[k] * cb[r];
                }
            };
            for (int t = 0; t < T - 1; ++t) {
                const In nxt = load_in(t + 1);
                const V* wn = in.wn;
                const V AnT = mix_row<V, K>(wn, AKT);
                // what the RTS adjoint needs from the stored state and the inputs alone
                const auto locJ = env.at_mat(t, SL::M_J);
                const auto locPi = env.at_mat(t, SL::M_PINV);
                env.st_mat(env.scr(4), AnT);                           // read back as rows: X An
                const V Pf = env.ld_mat(env.at_mat(t, SL::M_PFILT));
                const V dc = env.ld_vec_c(env.at_vec(t, SL::V_D));
                const V J = env.ld_mat(locJ), JT = env.ld_mat_t(locJ);
                const V Pi = env.ld_mat(locPi);
                const Q4<V> Pi4 = env.ld_rows4(locPi);
                env.st_mat_t(env.scr(1), mm4(J, env.ld_rows4(env.at_mat(t, SL::M_D))));   // J D
                const V WTs = skew(mm4(Pi, env.ld_rows4(env.scr(4))));   // Pn^-1 An = W^T
                const V JTs = skew(JT), Pfs = skew(Pf);
                const Q4<V> Jc4 = env.ld_cols4(locJ);
                V AbT, z1c, z1r, unr, unc;
                terms_adjoint(t, std::false_type{}, AbT, z1c, z1r, unr, unc);
                // steps below the burn-in pass no gradient on to the filter's moments
                const R keep = (unsigned)t < burn ? R(0) : R(1);
                msbc = keep * msbc;
                msbr = keep * msbr;
                Psb = keep * Psb;
                // adjoint of the RTS step t:  J = Pf W, W = An^T Pn^-1, with Z = Jb Pn^-1
                const V Hs = Psb + tr(Psb);
                const V H = sym ? half * Hs : Psb;
                const V Jb = msbr * dc + mm4(Hs, env.ld_rows4(env.scr(1)));   // msb d^T + (H + H^T) J D
                const V dbc = csum(J * msbr), dbr = qsum(JT * msbc);   // J^T msb
                const V Db = mm4(JTs, rots(mm4(H, Jc4)));              // J^T (H J)
                const V Z = mm4(Jb, Pi4);
                const V Pfb = H + mm4(Z, env.ld_rows4(env.scr(4)));    // + Jb W^T = Z An
                const V PZ = mm4(Pfs, rots(Z));                        // Pf Z = (Pn^-1 Wb^T)^T
                AbT += PZ;
                const V Pnb = -mm4(WTs, rots(PZ)) - Db;                // -Pn^-1 (An Pf Jb) Pn^-1 - Db
                env.st_vec(env.at_vec(t, SL::V_BM), msbc);
                env.st_mat(env.at_mat(t, SL::M_BP), Pfb);
                env.st_vec(env.at_vec(t, SL::V_BMN), -dbc);
                env.st_mat(env.at_mat(t, SL::M_BPN), Pnb);
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
            {
                // the last step: smoothed == filtered there (no burn-in gate)
                V AbT, z1c, z1r, unr, unc;
                terms_adjoint(T - 1, std::true_type{}, AbT, z1c, z1r, unr, unc);
                env.st_vec(env.at_vec(T - 1, SL::V_BM), msbc);
                env.st_mat(env.at_mat(T - 1, SL::M_BP), Psb);
                env.st_vec(env.at_vec(T - 1, SL::V_BMN), zero);
                env.st_mat(env.at_mat(T - 1, SL::M_BPN), zero);
            }
            // covariance gradients of the Gaussian terms: w/2 (sum u u^T - count Sigma^-1)
            KV_UNROLL
            for (int r = 0; r < P; ++r) {
                KV_UNROLL
                for (int c = 0; c < P; ++c)
                    acc.Rs[r][c] += (wo * half) * (suuR[r][c] - R(T) * Ri[r][c]);
            }
            acc.Q += (wt * half) * (suuQ - R(T - 1) * Qi);
        }
        env.fence();

        // ---- phase FA: filter adjoint, t = T-1..0 ----------------------------------------------
        if (KR_STOP_AFTER > 3) {
            V mpbc = zero, mpbr = zero, Ppb = zero;
            V dwn[K];
            KV_UNROLL
            for (int k = 0; k < K; ++k) dwn[k] = zero;
            // inputs of step t: weights of step t, observation, mask, and what phase SA left
            // in d_alpha[t+1] and d_a[t]
            struct In {
                V w[K], a[P], mu, da_old[K], a_old[P];
            };
            auto load_in = [&](int t) {
                In x;
                const long tb = (long)t * B + b;
                weights(tb, x.w);
                observation(tb, x.a);
                x.mu = f.mask ? env.scalar(f.mask + tb) : V(R(1));
                KV_UNROLL
                for (int k = 0; k < K; ++k)
                    x.da_old[k] = t + 1 < T ? env.scalar(g.d_alpha + (tb + B) * K + k) : zero;
                KV_UNROLL
                for (int r = 0; r < P; ++r) x.a_old[r] = env.scalar(g.d_a + tb * P + r);
                return x;
            };
            V wn[K];
            weights((long)T * B + b, wn);
            In in = load_in(T - 1);
            // the prediction step 0 starts from sits in two scratch slots, so that every step
            // reads its prediction the same way
            env.st_mat(env.scr(2), P0);
            env.st_vec(env.scr(3), m0c);
            env.fence();
            for (int t = T - 1; t >= 0; --t) {
                const long tb = (long)t * B + b;
                const In nxt = load_in(clamp_t(t - 1));
                const V* w = in.w;
                V cc[P], cr[P];
                mix_rows<V, P, K>(w, CKc, cc);
                mix_rows<V, P, K>(w, CKr, cr);
                const V An = mix_row<V, K>(wn, AK), AnT = mix_row<V, K>(wn, AKT);
                env.st_mat(env.scr(4), AnT);          // read back as rows: X An
                const V mpc = env.ld_vec_c(t > 0 ? env.at_vec(t - 1, SL::V_MN) : env.scr(3));
                const V Pp = env.ld_mat(t > 0 ? env.at_mat(t - 1, SL::M_PNEXT) : env.scr(2));
                const V mc = env.ld_vec_c(env.at_vec(t, SL::V_M));
                // 1. gradients of the step's four outputs
                V mbc = env.ld_vec_c(env.at_vec(t, SL::V_BM)), mbr = env.ld_vec_r(env.at_vec(t, SL::V_BM));
                V Pb = env.ld_mat(env.at_mat(t, SL::M_BP));
                mpbc += env.ld_vec_c(env.at_vec(t, SL::V_BMN));
                mpbr += env.ld_vec_r(env.at_vec(t, SL::V_BMN));
                Ppb += env.ld_mat(env.at_mat(t, SL::M_BPN));
                {
                    // steps below the burn-in pass no gradient on
                    const R keep = (unsigned)t < burn ? R(0) : R(1);
                    mbc = keep * mbc;
                    mbr = keep * mbr;
                    mpbc = keep * mpbc;
                    mpbr = keep * mpbr;
                    Pb = keep * Pb;
                    Ppb = keep * Ppb;
                }
                // 2. time update  m' = An m,  P' = sym(An Pf An^T + Q)
                {
                    const V Hs = Ppb + tr(Ppb);
                    const V H = sym ? half * Hs : Ppb;
                    env.st_mat_t(env.scr(0), mm4(An, env.ld_rows4(env.at_mat(t, SL::M_PFILT))));   // An Pf
                    const V Ab = mpbr * mc + mm4(Hs, env.ld_rows4(env.scr(0)));   // mpb m^T + (H + H^T) An Pf
                    mbc += csum(An * mpbr);                           // An^T mpb
                    mbr += qsum(AnT * mpbc);
                    Pb += mm4(mix_row<V, K>(wn, AKTs), rots(mm4(H, env.ld_rows4(env.scr(4)))));   // An^T (H An)
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
                const Update u = update_terms_row(cc, cr, mpc, Pp, in.a, 0);
                const V H2s = Pb + tr(Pb);
                const V H2 = sym ? half * H2s : Pb;
                // (mask weight: 1 without a mask)
                const V mobr = in.mu * mbr, Pob = in.mu * H2;
                const V PobT = sym ? Pob : in.mu * (H2s - Pb);
                mpbc = mbc;          // blend + identity term of m_obs
                mpbr = mbr;
                Ppb = H2;
                // m_obs = mp + sum_r g_r r_r,  P_obs = Pp - sum_r g_r (row r of M):
                V gbr[P], rb[P], cb[P], mkc[P], mkr[P], fbr[P];
                KV_UNROLL
                for (int r = 0; r < P; ++r) {
                    gbr[r] = mobr * u.r[r] - qsum(Pob * u.fc[r]);      // gradient w.r.t. g_r
                    rb[r] = csum(u.gr[r] * mobr);                      // g_r . mob
                    env.put_scalar(g.d_a + tb * P + r, in.a_old[r] + rb[r]);
                    cb[r] = -(rb[r] * mpc);
                    mpbc = mpbc - rb[r] * cc[r];                       // C^T rb
                    mpbr = mpbr - rb[r] * cr[r];
                    mkc[r] = -csum(Pob * u.gr[r]);                     // gradient w.r.t. row r of M:
                    mkr[r] = -qsum(PobT * u.gc[r]);                    // -Pob^T g_r, both forms
                    cb[r] = cb[r] + csum(Pp * mkr[r]);                 // M = C Pp
                    Ppb = Ppb + cr[r] * mkc[r];
                }
                // G = F Si:  Fb = Gb Si^T,  Sb = -Si^T (F^T Gb) Si^T = -G^T Fb
                V Sb[P][P];
                KV_UNROLL
                for (int c = 0; c < P; ++c) {
                    fbr[c] = gbr[0] * u.Si[c][0];
                    KV_UNROLL
                    for (int r = 1; r < P; ++r) fbr[c] = fbr[c] + gbr[r] * u.Si[c][r];
                }
                KV_UNROLL
                for (int r = 0; r < P; ++r) {
                    KV_UNROLL
                    for (int c = 0; c < P; ++c) Sb[r][c] = -csum(u.gr[r] * fbr[c]);
                }
                // F = Pp C^T;  S = C Pp C^T + R
                KV_UNROLL
                for (int r = 0; r < P; ++r) {
                    Ppb = Ppb + fbr[r] * cc[r];
                    cb[r] = cb[r] + csum(Pp * fbr[r]);
                    V sc = Sb[r][0] * cc[0];
                    KV_UNROLL
                    for (int c = 0; c < P; ++c) {
                        cb[r] = cb[r] + (Sb[r][c] + Sb[c][r]) * u.fc[c];   // Sb C Pp^T + Sb^T C Pp
                        if (c > 0) sc = sc + Sb[r][c] * cc[c];
                        acc.Rs[r][c] += Sb[r][c];
                    }
                    Ppb = Ppb + cr[r] * sc;                                // C^T Sb C
                }
                // 4. C-part of d_alpha[t]
                KV_UNROLL
                for (int k = 0; k < K; ++k) {
                    dwn[k] = qsum(dotp(cb, CKc[k]));
                    KV_UNROLL
                    for (int r = 0; r < P; ++r) acc.CKc[k][r] += w[k] * cb[r];
                }
                KV_UNROLL
                for (int k = 0; k < K; ++k) wn[k] = w[k];
                in = nxt;
                env.fence();
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
