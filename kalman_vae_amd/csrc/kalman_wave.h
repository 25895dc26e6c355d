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
// The workgroup is a single wavefront: its LDS instructions execute in program
// order, so a later ds_read of any lane sees an earlier ds_write of any other
// lane without waiting.  What is needed between two cooperative operations is
// only that the COMPILER keeps their memory accesses in order (a wavefront-scope
// fence emits no instruction) -- no s_barrier and, unlike __syncthreads(), no
// wait for outstanding global stores.
#define KW_SYNC()                                               \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
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

// LDS matrices are row-major with a leading dimension of cols + 1: a column
// walk (transposed operand, symmetrisation, triangular solves) then strides by
// an odd number of words and is free of bank conflicts; with ld = cols = 32
// every element of a column sits in the same bank.  Vectors (cols == 1) are
// contiguous.  Element-wise operations run over the padded storage (the pad
// words are never read by anything that matters).
constexpr int ld(int cols) { return cols > 1 ? cols + 1 : 1; }
constexpr int msize(int rows, int cols) { return rows * ld(cols); }
// padded offset of logical element e of a (rows x COLS) matrix
template <int COLS>
KV_HD int pad_index(int e) {
    return (e / COLS) * ld(COLS) + (e % COLS);
}

// ---------------------------------------------------------------------------
// cooperative operations on LDS-resident matrices
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
// A is stored M x Kd (Kd x M when TA), B is stored Kd x N (N x Kd when TB).
template <typename R, int M, int Kd, int N, bool TA, bool TB, int MODE>
KV_HD void gemm_valu(const R* A, const R* Bm, R* C) {
    constexpr int lda = ld(TA ? M : Kd), ldb = ld(TB ? Kd : N), ldc = ld(N);
    KW_FOR(e, M * N) {
        const int i = e / N, j = e % N;
        R acc = R(0);
#pragma unroll 8
        for (int k = 0; k < Kd; ++k) {
            const R a = TA ? A[k * lda + i] : A[i * lda + k];
            const R b = TB ? Bm[j * ldb + k] : Bm[k * ldb + j];
            acc += a * b;
        }
        R* c = C + i * ldc + j;
        if (MODE == 0) *c = acc;
        if (MODE == 1) *c += acc;
        if (MODE == 2) *c -= acc;
    }
    KW_SYNC();
}

#if KW_DEVICE
// fp32 products with M, N <= 32 on the matrix core: v_mfma_f32_32x32x2_f32
// (exact fp32 multiply-add chain in k order).  Lane l supplies A[l&31][2kk+(l>>5)]
// and B[2kk+(l>>5)][l&31]; rows/columns beyond M/N and k beyond Kd are fed as
// zeros.  Accumulator register r of lane l holds C[(r&3)+8(r>>2)+4(l>>5)][l&31].
template <int M, int Kd, int N, bool TA, bool TB, int MODE>
__device__ __forceinline__ void gemm_mfma(const float* A, const float* Bm, float* C) {
    static_assert(M <= 32 && N <= 32, "one 32x32 accumulator tile");
    constexpr int lda = ld(TA ? M : Kd), ldb = ld(TB ? Kd : N), ldc = ld(N);
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int i = threadIdx.x & 31, h = threadIdx.x >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < (Kd + 1) / 2; ++kk) {
        const int k = 2 * kk + h;
        float a = 0.f, b = 0.f;
        if (i < M && k < Kd) a = TA ? A[k * lda + i] : A[i * lda + k];
        if (i < N && k < Kd) b = TB ? Bm[i * ldb + k] : Bm[k * ldb + i];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M && i < N) {
            float* c = C + row * ldc + i;
            if (MODE == 0) *c = acc[r];
            if (MODE == 1) *c += acc[r];
            if (MODE == 2) *c -= acc[r];
        }
    }
    KW_SYNC();
}
#endif

template <typename R, int M, int Kd, int N, bool TA, bool TB, int MODE>
KV_HD void gemm(const R* A, const R* Bm, R* C) {
#if KW_DEVICE
    if constexpr (sizeof(R) == 4 && M * N >= 256 && M <= 32 && N <= 32 && Kd >= 8) {
        gemm_mfma<M, Kd, N, TA, TB, MODE>(A, Bm, C);
        return;
    }
#endif
    gemm_valu<R, M, Kd, N, TA, TB, MODE>(A, Bm, C);
}

template <typename R, int M, int N, bool TA, int MODE>
KV_HD void gemv(const R* A, const R* x, R* y) {
    gemm<R, M, N, 1, TA, false, MODE>(A, x, y);
}

// C (M x N) += s x y^T
template <typename R, int M, int N>
KV_HD void outer_add(R s, const R* x, const R* y, R* C) {
    KW_FOR(e, M * N) C[pad_index<N>(e)] += s * x[e / N] * y[e % N];
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
            const R v = (X[i * ld(N) + j] + X[j * ld(N) + i]) / R(2);
            X[i * ld(N) + j] = v;
            X[j * ld(N) + i] = v;
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

// matrix variants: global memory is dense row-major, LDS is padded
template <typename R, int ROWS, int COLS>
KV_HD void gload_mat(const R* p, R* x) {
    KW_FOR(e, ROWS * COLS) x[pad_index<COLS>(e)] = p[e];
    KW_SYNC();
}
template <typename R, int ROWS, int COLS>
KV_HD void gload_mat_or_zero(const R* base, long off, R* x) {
    KW_FOR(e, ROWS * COLS) x[pad_index<COLS>(e)] = base ? base[off + e] : R(0);
    KW_SYNC();
}
template <typename R, int ROWS, int COLS>
KV_HD void gload_mat_add(const R* base, long off, R* x) {
    if (base) {
        KW_FOR(e, ROWS * COLS) x[pad_index<COLS>(e)] += base[off + e];
    }
    KW_SYNC();
}
template <typename R, int ROWS, int COLS>
KV_HD void gstore_mat(R* p, const R* x) {
    KW_FOR(e, ROWS * COLS) p[e] = x[pad_index<COLS>(e)];
    KW_SYNC();
}

template <typename R, int K, int ROWS, int COLS>
KV_HD void mix(const R* w, const R* bases, R* out) {
    constexpr int Len = ROWS * COLS;
    KW_FOR(e, Len) {
        R acc = R(0);
#pragma unroll
        for (int k = 0; k < K; ++k) acc += w[k] * bases[k * Len + e];
        out[pad_index<COLS>(e)] = acc;
    }
    KW_SYNC();
}

// Right-looking Cholesky, lower factor; S is destroyed (used as work space).
// Only the lower triangle of S is read.  Returns false on a pivot <= 0.
// Generic form (any real type, also the host build).
template <typename R, int N>
KV_HD bool cholesky_generic(R* S, R* L) {
    constexpr int l = ld(N);
    bool ok = true;
    for (int j = 0; j < N; ++j) {
        R d = S[j * l + j];  // same address in every lane
        if (!(d > R(0))) {
            ok = false;
            d = R(1);
        }
        const R ljj = sqrt(d);
        const R inv = R(1) / ljj;
        KW_SYNC();
        KW_FOR(i, N) {
            if (i < j) L[i * l + j] = R(0);
            if (i == j) L[i * l + j] = ljj;
            if (i > j) L[i * l + j] = S[i * l + j] * inv;
        }
        KW_SYNC();
        // trailing update, one lane per row: S[i][k] -= L[i][j] L[k][j], i >= k > j
        KW_FOR(i, N) {
            if (i > j) {
                const R lij = L[i * l + j];
                for (int k = j + 1; k <= i; ++k) S[i * l + k] -= lij * L[k * l + j];
            }
        }
        KW_SYNC();
    }
    return ok;
}

// Inverse of a lower-triangular matrix; lane j builds column j.
template <typename R, int N>
KV_HD void tri_lower_inverse_generic(const R* L, R* Li) {
    constexpr int l = ld(N);
    KW_FOR(j, N) {
        for (int i = 0; i < N; ++i) {
            if (i < j) {
                Li[i * l + j] = R(0);
            } else if (i == j) {
                Li[i * l + j] = R(1) / L[i * l + i];
            } else {
                R s = R(0);
                for (int k = j; k < i; ++k) s += L[i * l + k] * Li[k * l + j];
                Li[i * l + j] = -s / L[i * l + i];
            }
        }
    }
    KW_SYNC();
}

#if KW_DEVICE
// fp32, register-resident: lane i keeps row i of the factor in N registers and
// the other rows' elements arrive through v_readlane (a scalar operand of the
// FMA), so the factorisation needs no LDS traffic and no barrier until the
// result is written back.  The loops are fully unrolled (register arrays must be
// indexed with constants); the functions are noinline so that each size exists
// once per kernel.
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// in: S (lower triangle read); out: L and, if Li != nullptr, L^-1 (both N x N, padded)
template <int N>
__device__ __noinline__ bool cholesky_inverse_f32(const float* S, float* L, float* Li) {
    static_assert(N <= 64, "one row per lane");
    constexpr int l = ld(N);
    const int lane = threadIdx.x;
    const int i = lane < N ? lane : N - 1;
    float row[N];
#pragma unroll
    for (int k = 0; k < N; ++k) row[k] = S[i * l + k];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float s = row[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= row[k] * lane_bcast(row[k], j);
        float d = lane_bcast(s, j);
        if (!(d > 0.f)) {
            ok = false;
            d = 1.f;
        }
        const float ljj = sqrtf(d);
        const float inv = 1.f / ljj;
        row[j] = (i == j) ? ljj : (i > j ? s * inv : 0.f);
    }
    if (lane < N) {
#pragma unroll
        for (int k = 0; k < N; ++k) L[lane * l + k] = row[k];
    }
    if (Li) {
        // own diagonal element without a run-time register index
        float diag = 1.f;
#pragma unroll
        for (int k = 0; k < N; ++k) diag = (i == k) ? row[k] : diag;
        const float rdiag = 1.f / diag;
        float x[N];  // column `lane` of L^-1
#pragma unroll
        for (int r = 0; r < N; ++r) {
            float acc = (lane == r) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < r; ++k) acc -= lane_bcast(row[k], r) * x[k];
            x[r] = acc * lane_bcast(rdiag, r);
        }
        if (lane < N) {
#pragma unroll
            for (int r = 0; r < N; ++r) Li[r * l + lane] = x[r];
        }
    }
    KW_SYNC();
    return ok;
}

template <int N>
__device__ __noinline__ void tri_lower_inverse_f32(const float* L, float* Li) {
    constexpr int l = ld(N);
    const int lane = threadIdx.x;
    const int i = lane < N ? lane : N - 1;
    float row[N];
#pragma unroll
    for (int k = 0; k < N; ++k) row[k] = L[i * l + k];
    float diag = 1.f;
#pragma unroll
    for (int k = 0; k < N; ++k) diag = (i == k) ? row[k] : diag;
    const float rdiag = 1.f / diag;
    float x[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        float acc = (lane == r) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < r; ++k) acc -= lane_bcast(row[k], r) * x[k];
        x[r] = acc * lane_bcast(rdiag, r);
    }
    if (lane < N) {
#pragma unroll
        for (int r = 0; r < N; ++r) Li[r * l + lane] = x[r];
    }
    KW_SYNC();
}
#endif

// S may be destroyed.
template <typename R, int N>
KV_HD bool cholesky(R* S, R* L) {
#if KW_DEVICE
    if constexpr (sizeof(R) == 4) return cholesky_inverse_f32<N>(S, L, nullptr);
#endif
    return cholesky_generic<R, N>(S, L);
}

template <typename R, int N>
KV_HD void tri_lower_inverse(const R* L, R* Li) {
#if KW_DEVICE
    if constexpr (sizeof(R) == 4) {
        tri_lower_inverse_f32<N>(L, Li);
        return;
    }
#endif
    tri_lower_inverse_generic<R, N>(L, Li);
}

// Si = S^-1 through the Cholesky factor; S may be destroyed; L and Li are outputs.
template <typename R, int N>
KV_HD bool spd_inverse(R* S, R* Si, R* L, R* Li) {
    bool ok;
#if KW_DEVICE
    if constexpr (sizeof(R) == 4) {
        ok = cholesky_inverse_f32<N>(S, L, Li);
    } else
#endif
    {
        ok = cholesky_generic<R, N>(S, L);
        tri_lower_inverse_generic<R, N>(L, Li);
    }
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
            for (int k = i; k < N; ++k) acc += L[k * ld(N) + i] * Lbar[k * ld(N) + j];
            if (i == j) acc *= R(0.5);
        }
        X[pad_index<N>(e)] = acc;
    }
    KW_SYNC();
    gemm<R, N, N, N, true, false, 0>(Li, X, Y);   // L^-T X
    gemm<R, N, N, N, false, false, 0>(Y, Li, X);  // ... L^-1
    KW_FOR(e, N * N) {
        const int i = e / N, j = e % N;
        Sacc[pad_index<N>(e)] += (X[i * ld(N) + j] + X[j * ld(N) + i]) * R(0.5);
    }
    KW_SYNC();
}

// sum_i log L[i][i]
template <typename R, int N>
KV_HD R half_logdet(const R* L) {
    R acc = R(0);
    KW_FOR(i, N) acc += log(L[i * ld(N) + i]);
    return wsum<R>(acc);
}

// Register-resident accumulator of LEN logical values distributed over the lanes.
template <typename R, int LEN>
struct Acc {
    static constexpr int PER = (LEN + KW_LANES - 1) / KW_LANES;
    R v[PER];
    KV_HD void zero() {
        KV_UNROLL
        for (int q = 0; q < PER; ++q) v[q] = R(0);
    }
    // this += s * x, x a (LEN/COLS x COLS) matrix in LDS
    template <int COLS>
    KV_HD void axpy(R s, const R* x) {
#if KW_DEVICE
        KV_UNROLL
        for (int q = 0; q < PER; ++q) {
            const int e = (int)threadIdx.x + KW_LANES * q;
            if (e < LEN) v[q] += s * x[pad_index<COLS>(e)];
        }
#else
        for (int e = 0; e < LEN; ++e) v[e] += s * x[pad_index<COLS>(e)];
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

// zero a whole scratch struct (pad words included) once per kernel
template <typename S, typename R>
KV_HD void clear_scratch(S& s) {
    R* w = reinterpret_cast<R*>(&s);
    KW_FOR(e, (int)(sizeof(S) / sizeof(R))) w[e] = R(0);
    KW_SYNC();
}

// ---------------------------------------------------------------------------
// Filter
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
struct FilterScratch {
    static constexpr int NN = msize(N, N), PN = msize(P, N), NP = msize(N, P), PP = msize(P, P);
    R Rm[PP], Q[NN], mp[N], Pp[NN], m[N], Pf[NN];
    R w[K], wn[K], a[P], C[PN], An[NN];
    R M[PN], S[PP], Si[PP], LS[PP], LiS[PP], F[NP], G[NP], r[P];
    R T1[NN];
};
template <typename R, int N, int P, int K>
struct FilterBwdScratch : FilterScratch<R, N, P, K> {
    using B0 = FilterScratch<R, N, P, K>;
    R mb[N], Pb[B0::NN], mpb[N], Ppb[B0::NN], Ab[B0::NN], Cb[B0::PN], T2[B0::NN];
    R mob[N], Pob[B0::NN], Gb[B0::NP], rb[P], Mb[B0::PN], Fb[B0::NP], Sib[B0::PP], Sb[B0::PP],
        T3[B0::PP], CPt[B0::PN], T4[B0::PN];
};

template <typename R, int N, int P, int K>
KV_HD bool update_terms(FilterScratch<R, N, P, K>& s) {
    gemm<R, P, N, N, false, false, 0>(s.C, s.Pp, s.M);
    vcopy<R, msize(P, P)>(s.Rm, s.S);
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
    constexpr int NN = msize(N, N);
    const long B = g.B;
    clear_scratch<FilterScratch<R, N, P, K>, R>(s);
    gload_mat<R, P, P>(g.Rm, s.Rm);
    gload_mat<R, N, N>(g.Q, s.Q);
    if (g.carry_per_sequence) {
        gload<R, N>(g.m_in + (long)b * N, s.mp);
        gload_mat<R, N, N>(g.P_in + (long)b * N * N, s.Pp);
    } else {
        gload<R, N>(g.m_in, s.mp);
        gload_mat<R, N, N>(g.P_in, s.Pp);
    }
    gload<R, K>(g.alpha + (long)b * K, s.w);
    if (g.As) {
        mix<R, K, N, N>(s.w, g.A_K, s.An);
        gstore_mat<R, N, N>(g.As + (long)b * N * N, s.An);
    }
    int st = 0;
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        gload<R, K>(g.alpha + (tb + B) * K, s.wn);
        gload<R, P>(g.a + tb * P, s.a);
        mix<R, K, P, N>(s.w, g.C_K, s.C);
        mix<R, K, N, N>(s.wn, g.A_K, s.An);
        if (g.Cs) gstore_mat<R, P, N>(g.Cs + tb * P * N, s.C);
        if (g.As) gstore_mat<R, N, N>(g.As + (tb + B) * N * N, s.An);
        if (!update_terms<R, N, P, K>(s)) st |= ST_NONPD_INNOVATION;

        vcopy<R, N>(s.mp, s.m);
        gemv<R, N, P, false, 1>(s.G, s.r, s.m);
        vcopy<R, NN>(s.Pp, s.Pf);
        gemm<R, N, P, N, false, false, 2>(s.G, s.M, s.Pf);
        if (g.mask) {
            const R mu = g.mask[tb];
            axpby<R, N>(mu, s.m, R(1) - mu, s.mp, s.m);
            axpby<R, NN>(mu, s.Pf, R(1) - mu, s.Pp, s.Pf);
        }
        if (g.symmetrize) symmetrize_inplace<R, N>(s.Pf);
        gemv<R, N, N, false, 0>(s.An, s.m, s.mp);
        gemm<R, N, N, N, false, false, 0>(s.An, s.Pf, s.T1);
        vcopy<R, NN>(s.Q, s.Pp);
        gemm<R, N, N, N, false, true, 1>(s.T1, s.An, s.Pp);
        if (g.symmetrize) symmetrize_inplace<R, N>(s.Pp);
        gstore<R, N>(g.m + tb * N, s.m);
        gstore_mat<R, N, N>(g.Pf + tb * N * N, s.Pf);
        gstore<R, N>(g.mn + tb * N, s.mp);
        gstore_mat<R, N, N>(g.Pn + tb * N * N, s.Pp);
        vcopy<R, K>(s.wn, s.w);
    }
    if (g.Cs) {
        mix<R, K, P, N>(s.w, g.C_K, s.C);
        gstore_mat<R, P, N>(g.Cs + ((long)g.T * B + b) * P * N, s.C);
    }
    raise_status(g.status, st);
}

// Sums over t of the noise-covariance gradients of one sequence.  The basis
// gradients and d_alpha are NOT accumulated here: the kernel stores the
// gradients w.r.t. the mixed matrices A_t, C_t (bar_As, bar_Cs) and a separate
// streaming pass contracts them with alpha and the bases (mix_adjoint_*), which
// keeps ~190 accumulator registers out of this kernel.
template <typename R, int N, int P, int K>
struct FilterGradAcc {
    static constexpr int LEN = N * N + P * P;
    Acc<R, N * N> dQ;
    Acc<R, P * P> dR;
    KV_HD void zero() {
        dQ.zero();
        dR.zero();
    }
    KV_HD void store(R* row) const {
        dQ.store(row);
        dR.store(row + N * N);
    }
};

template <typename R, int N, int P, int K>
KV_HD void filter_backward_sequence(const FilterBwdArgs<R>& g, int b,
                                    FilterBwdScratch<R, N, P, K>& s,
                                    FilterGradAcc<R, N, P, K>& acc) {
    constexpr int NN = msize(N, N), PP = msize(P, P), NP = msize(N, P);
    const FilterFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    clear_scratch<FilterBwdScratch<R, N, P, K>, R>(s);
    gload_mat<R, P, P>(f.Rm, s.Rm);
    {
        // C_T only feeds the returned mat_Cs[T]
        gload_mat_or_zero<R, P, N>(g.g_Cs, ((long)T * B + b) * P * N, s.Cb);
        gstore_mat<R, P, N>(g.bar_Cs + ((long)T * B + b) * P * N, s.Cb);
    }
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        gload<R, K>(f.alpha + tb * K, s.w);
        gload<R, K>(f.alpha + (tb + B) * K, s.wn);
        gload<R, P>(f.a + tb * P, s.a);
        mix<R, K, P, N>(s.w, f.C_K, s.C);
        mix<R, K, N, N>(s.wn, f.A_K, s.An);
        if (t > 0) {
            gload<R, N>(f.mn + (tb - B) * N, s.mp);
            gload_mat<R, N, N>(f.Pn + (tb - B) * N * N, s.Pp);
        } else if (f.carry_per_sequence) {
            gload<R, N>(f.m_in + (long)b * N, s.mp);
            gload_mat<R, N, N>(f.P_in + (long)b * N * N, s.Pp);
        } else {
            gload<R, N>(f.m_in, s.mp);
            gload_mat<R, N, N>(f.P_in, s.Pp);
        }
        gload<R, N>(f.m + tb * N, s.m);
        gload_mat<R, N, N>(f.Pf + tb * N * N, s.Pf);

        gload_or_zero<R, N>(g.g_m, tb * N, s.mb);
        gload_mat_or_zero<R, N, N>(g.g_P, tb * N * N, s.Pb);
        gload_add<R, N>(g.g_mn, tb * N, s.mpb);
        gload_mat_add<R, N, N>(g.g_Pn, tb * N * N, s.Ppb);
        if (t < g.burn_in) {
            vzero<R, N>(s.mb);
            vzero<R, NN>(s.Pb);
            vzero<R, N>(s.mpb);
            vzero<R, NN>(s.Ppb);
        }
        // time update adjoint
        gload_mat_or_zero<R, N, N>(g.g_As, (tb + B) * N * N, s.Ab);
        if (f.symmetrize) symmetrize_inplace<R, N>(s.Ppb);
        outer_add<R, N, N>(R(1), s.mpb, s.m, s.Ab);
        gemm<R, N, N, N, false, true, 0>(s.An, s.Pf, s.T1);
        gemm<R, N, N, N, false, false, 1>(s.Ppb, s.T1, s.Ab);
        gemm<R, N, N, N, false, false, 0>(s.An, s.Pf, s.T1);
        gemm<R, N, N, N, true, false, 1>(s.Ppb, s.T1, s.Ab);
        gemv<R, N, N, true, 1>(s.An, s.mpb, s.mb);
        gemm<R, N, N, N, true, false, 0>(s.An, s.Ppb, s.T2);
        gemm<R, N, N, N, false, false, 1>(s.T2, s.An, s.Pb);
        acc.dQ.template axpy<N>(R(1), s.Ppb);

        // measurement update adjoint
        update_terms<R, N, P, K>(s);
        const R mu = f.mask ? f.mask[tb] : R(1);
        const R nu = R(1) - mu;
        if (f.symmetrize) symmetrize_inplace<R, N>(s.Pb);
        gload_mat_or_zero<R, P, N>(g.g_Cs, tb * P * N, s.Cb);
        KW_FOR(i, N) {
            s.mob[i] = mu * s.mb[i];
            s.mpb[i] = nu * s.mb[i] + mu * s.mb[i];
        }
        KW_FOR(i, NN) {
            s.Pob[i] = mu * s.Pb[i];
            s.Ppb[i] = nu * s.Pb[i] + mu * s.Pb[i];
        }
        KW_SYNC();
        vzero<R, NP>(s.Gb);
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
        vscale<R, PP>(R(-1), s.Sb);
        gemm<R, N, P, N, false, false, 1>(s.Fb, s.C, s.Ppb);
        gemm<R, P, N, N, true, false, 1>(s.Fb, s.Pp, s.Cb);
        gemm<R, P, N, N, false, true, 0>(s.C, s.Pp, s.CPt);
        gemm<R, P, P, N, false, false, 1>(s.Sb, s.CPt, s.Cb);
        gemm<R, P, P, N, true, false, 1>(s.Sb, s.M, s.Cb);
        gemm<R, P, P, N, false, false, 0>(s.Sb, s.C, s.T4);
        gemm<R, N, P, N, true, false, 1>(s.C, s.T4, s.Ppb);
        acc.dR.template axpy<P>(R(1), s.Sb);

        // gradients w.r.t. the mixed matrices of this step
        gstore_mat<R, N, N>(g.bar_As + (tb + B) * N * N, s.Ab);
        gstore_mat<R, P, N>(g.bar_Cs + tb * P * N, s.Cb);
    }
    {
        // A_0 only feeds the returned mat_As[0]
        gload_mat_or_zero<R, N, N>(g.g_As, (long)b * N * N, s.Ab);
        gstore_mat<R, N, N>(g.bar_As + (long)b * N * N, s.Ab);
    }
    gstore<R, N>(g.d_m_in + (long)b * N, s.mpb);
    gstore_mat<R, N, N>(g.d_P_in + (long)b * N * N, s.Ppb);
}

// ---------------------------------------------------------------------------
// Smoother
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
struct SmoothScratch {
    static constexpr int NN = msize(N, N), PN = msize(P, N), PP = msize(P, P);
    R LR[PP], ms[N], Ps[NN], C[PN], m[N], Pf[NN], mn[N], Pn[NN], An[NN];
    R Pi[NN], L[NN], Li[NN], W[NN], J[NN], d[N], D[NN], T1[NN];
    R z[N], ar[P], e[N], eo[P];
};
template <typename R, int N, int P>
struct SmoothBwdScratch : SmoothScratch<R, N, P> {
    using B0 = SmoothScratch<R, N, P>;
    R ms1[N], Ps1[B0::NN], msb[N], Psb[B0::NN], Cb[B0::PN], zb[N], ab[P], Lb[B0::NN], X[B0::NN],
        Y[B0::NN], Pfb[B0::NN], Jb[B0::NN], db[N], Db[B0::NN], Wb[B0::NN], Anb[B0::NN],
        Pib[B0::NN], Pnb[B0::NN];
};

template <typename R, int N, int P>
KV_HD bool smooth_emit(const SmoothFwdArgs<R>& g, long tb, SmoothScratch<R, N, P>& s) {
    bool ok = true;
    vcopy<R, N>(s.ms, s.z);
    if (g.eps_s) {
        vcopy<R, msize(N, N)>(s.Ps, s.T1);
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

// J = Pf An^T Pn^-1, d = ms_next - mn, D = Ps_next - Pn
template <typename R, int N, int P>
KV_HD bool smooth_gain(SmoothScratch<R, N, P>& s, const R* ms_next, const R* Ps_next) {
    constexpr int NN = msize(N, N);
    vcopy<R, NN>(s.Pn, s.T1);
    const bool ok = spd_inverse<R, N>(s.T1, s.Pi, s.L, s.Li);
    gemm<R, N, N, N, true, false, 0>(s.An, s.Pi, s.W);
    gemm<R, N, N, N, false, false, 0>(s.Pf, s.W, s.J);
    axpby<R, N>(R(1), ms_next, R(-1), s.mn, s.d);
    axpby<R, NN>(R(1), Ps_next, R(-1), s.Pn, s.D);
    return ok;
}

template <typename R, int N, int P>
KV_HD void smooth_forward_sequence(const SmoothFwdArgs<R>& g, int b, SmoothScratch<R, N, P>& s) {
    constexpr int NN = msize(N, N);
    const long B = g.B;
    const int T = g.T;
    clear_scratch<SmoothScratch<R, N, P>, R>(s);
    if (g.eps_o) gload_mat<R, P, P>(g.LR, s.LR);
    int st = 0;
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(g.m + tb * N, s.ms);
        gload_mat<R, N, N>(g.Pf + tb * N * N, s.Ps);
        gload_mat<R, P, N>(g.Cs + (tb + B) * P * N, s.C);
        gstore<R, N>(g.ms + tb * N, s.ms);
        gstore_mat<R, N, N>(g.Ps + tb * N * N, s.Ps);
        if (!smooth_emit<R, N, P>(g, tb, s)) st |= ST_NONPD_SMOOTHED;
    }
    for (int t = T - 2; t >= 0; --t) {
        const long tb = (long)t * B + b;
        gload<R, N>(g.m + tb * N, s.m);
        gload_mat<R, N, N>(g.Pf + tb * N * N, s.Pf);
        gload<R, N>(g.mn + tb * N, s.mn);
        gload_mat<R, N, N>(g.Pn + tb * N * N, s.Pn);
        gload_mat<R, N, N>(g.As + (tb + B) * N * N, s.An);
        gload_mat<R, P, N>(g.Cs + tb * P * N, s.C);
        if (!smooth_gain<R, N, P>(s, s.ms, s.Ps)) st |= ST_NONPD_PREDICTED;
        vcopy<R, N>(s.m, s.ms);
        gemv<R, N, N, false, 1>(s.J, s.d, s.ms);
        gemm<R, N, N, N, false, false, 0>(s.J, s.D, s.T1);
        vcopy<R, NN>(s.Pf, s.Ps);
        gemm<R, N, N, N, false, true, 1>(s.T1, s.J, s.Ps);
        if (g.symmetrize) symmetrize_inplace<R, N>(s.Ps);
        gstore<R, N>(g.ms + tb * N, s.ms);
        gstore_mat<R, N, N>(g.Ps + tb * N * N, s.Ps);
        if (!smooth_emit<R, N, P>(g, tb, s)) st |= ST_NONPD_SMOOTHED;
    }
    raise_status(g.status, st);
}

// adjoint of the emission at pair tb for the smoothed (ms, Ps) in s.ms/s.Ps and C in s.C:
// adds into s.msb / s.Psb, writes s.Cb.
template <typename R, int N, int P>
KV_HD void smooth_emit_adjoint(const SmoothBwdArgs<R>& g, long tb, SmoothBwdScratch<R, N, P>& s,
                               Acc<R, P * P>& acc_LR) {
    constexpr int NN = msize(N, N);
    const SmoothFwdArgs<R>& f = g.f;
    gload_or_zero<R, N>(g.g_zs, tb * N, s.zb);
    gload_or_zero<R, P>(g.g_ares, tb * P, s.ab);
    vzero<R, msize(P, N)>(s.Cb);
    if (!g.g_ares && !g.g_zs) return;
    if (f.eps_s) {
        vcopy<R, NN>(s.Ps, s.T1);
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
            vzero<R, msize(P, P)>(s.X);
            outer_add<R, P, P>(R(1), s.ab, s.eo, s.X);
            acc_LR.template axpy<P>(R(1), s.X);
        }
    }
    vadd<R, N>(s.zb, s.msb);
    if (f.eps_s) {
        tri_lower_inverse<R, N>(s.L, s.Li);
        vzero<R, NN>(s.Lb);
        outer_add<R, N, N>(R(1), s.zb, s.e, s.Lb);
        cholesky_adjoint_add<R, N>(s.L, s.Li, s.Lb, s.X, s.Y, s.Psb);
    }
}

template <typename R, int N, int P>
KV_HD void smooth_backward_sequence(const SmoothBwdArgs<R>& g, int b,
                                    SmoothBwdScratch<R, N, P>& s, Acc<R, P * P>& acc_LR) {
    constexpr int NN = msize(N, N);
    const SmoothFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    clear_scratch<SmoothBwdScratch<R, N, P>, R>(s);
    if (f.eps_o) gload_mat<R, P, P>(f.LR, s.LR);
    gstore_zero<R, N * N>(g.d_As + (long)b * N * N);
    gstore_zero<R, N * N>(g.d_As + ((long)T * B + b) * N * N);
    gstore_zero<R, P * N>(g.d_Cs + ((long)(T - 1) * B + b) * P * N);
    gstore_zero<R, N>(g.d_mn + ((long)(T - 1) * B + b) * N);
    gstore_zero<R, N * N>(g.d_Pn + ((long)(T - 1) * B + b) * N * N);
    for (int t = 0; t < T - 1; ++t) {
        const long tb = (long)t * B + b;
        gload<R, N>(f.m + tb * N, s.m);
        gload_mat<R, N, N>(f.Pf + tb * N * N, s.Pf);
        gload<R, N>(f.mn + tb * N, s.mn);
        gload_mat<R, N, N>(f.Pn + tb * N * N, s.Pn);
        gload_mat<R, N, N>(f.As + (tb + B) * N * N, s.An);
        gload_mat<R, P, N>(f.Cs + tb * P * N, s.C);
        gload<R, N>(f.ms + (tb + B) * N, s.ms1);
        gload_mat<R, N, N>(f.Ps + (tb + B) * N * N, s.Ps1);
        gload<R, N>(f.ms + tb * N, s.ms);
        gload_mat<R, N, N>(f.Ps + tb * N * N, s.Ps);
        gload_add<R, N>(g.g_ms, tb * N, s.msb);
        gload_mat_add<R, N, N>(g.g_Ps, tb * N * N, s.Psb);
        smooth_emit_adjoint<R, N, P>(g, tb, s, acc_LR);
        gstore_mat<R, P, N>(g.d_Cs + tb * P * N, s.Cb);
        if (t < g.burn_in) {
            vzero<R, N>(s.msb);
            vzero<R, NN>(s.Psb);
        }
        smooth_gain<R, N, P>(s, s.ms1, s.Ps1);
        if (f.symmetrize) symmetrize_inplace<R, N>(s.Psb);
        gstore<R, N>(g.d_m + tb * N, s.msb);
        vcopy<R, NN>(s.Psb, s.Pfb);
        vzero<R, NN>(s.Jb);
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
        axpby<R, NN>(R(-1), s.Pnb, R(-1), s.Db, s.Pnb);
        gstore_mat<R, N, N>(g.d_P + tb * N * N, s.Pfb);
        KW_FOR(i, N) g.d_mn[tb * N + i] = -s.db[i];
        gstore_mat<R, N, N>(g.d_Pn + tb * N * N, s.Pnb);
        gstore_mat<R, N, N>(g.d_As + (tb + B) * N * N, s.Anb);
        vcopy<R, N>(s.db, s.msb);
        vcopy<R, NN>(s.Db, s.Psb);
    }
    {
        const long tb = (long)(T - 1) * B + b;
        gload<R, N>(f.ms + tb * N, s.ms);
        gload_mat<R, N, N>(f.Ps + tb * N * N, s.Ps);
        gload_mat<R, P, N>(f.Cs + (tb + B) * P * N, s.C);
        gload_add<R, N>(g.g_ms, tb * N, s.msb);
        gload_mat_add<R, N, N>(g.g_Ps, tb * N * N, s.Psb);
        smooth_emit_adjoint<R, N, P>(g, tb, s, acc_LR);
        gstore_mat<R, P, N>(g.d_Cs + (tb + B) * P * N, s.Cb);
        gstore<R, N>(g.d_m + tb * N, s.msb);
        gstore_mat<R, N, N>(g.d_P + tb * N * N, s.Psb);
    }
}

// ---------------------------------------------------------------------------
// ELBO terms, one wavefront per sequence, sequential in t.
// Forward runs t = 0..T-1 keeping z_{t-1}; backward runs t = T-1..0 keeping
// the next pair's transition gradient direction A_{t+1}^T u_{t+1}.
// ---------------------------------------------------------------------------
template <typename R, int N, int P>
struct ElboScratch {
    static constexpr int NN = msize(N, N), PN = msize(P, N), PP = msize(P, P);
    R Q[NN], LQ[NN], LiQ[NN], Rm[PP], LRm[PP], LiR[PP];
    R P0[NN], L0[NN], Li0[NN], m0[N];
    R Ps[NN], L[NN], Li[NN], z[N], zp[N], e[N], C[PN], A[NN], r[N], ro[P], y[N];
    R T1[NN];
};
template <typename R, int N, int P>
struct ElboBwdScratch : ElboScratch<R, N, P> {
    using B0 = ElboScratch<R, N, P>;
    R zn_back[N], zb[N], Ab[B0::NN], Cb[B0::PN], Lb[B0::NN], X[B0::NN], Y[B0::NN], Psb[B0::NN],
        SiQ[B0::NN], SiR[B0::PP], Si0[B0::NN], G1[B0::NN], G2[B0::PP];
};

// s.z = ms + chol(Ps) eps at pair tb; leaves L in s.L, eps in s.e
template <typename R, int N, int P>
KV_HD bool elbo_sample(const ElboFwdArgs<R>& g, long tb, ElboScratch<R, N, P>& s) {
    gload_mat<R, N, N>(g.Ps + tb * N * N, s.T1);
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
    constexpr int NN = msize(N, N), PP = msize(P, P);
    int st = 0;
    gload_mat<R, N, N>(g.Q, s.Q);
    vcopy<R, NN>(s.Q, s.T1);
    if (!cholesky<R, N>(s.T1, s.LQ)) st |= ST_NONPD_NOISE;
    tri_lower_inverse<R, N>(s.LQ, s.LiQ);
    gload_mat<R, P, P>(g.Rm, s.Rm);
    vcopy<R, PP>(s.Rm, s.T1);
    if (!cholesky<R, P>(s.T1, s.LRm)) st |= ST_NONPD_NOISE;
    tri_lower_inverse<R, P>(s.LRm, s.LiR);
    gload_mat<R, N, N>(g.P0, s.P0);
    vcopy<R, NN>(s.P0, s.T1);
    if (!cholesky<R, N>(s.T1, s.L0)) st |= ST_NONPD_NOISE;
    tri_lower_inverse<R, N>(s.L0, s.Li0);
    gload<R, N>(g.m0, s.m0);
    return st;
}

template <typename R, int N, int P>
KV_HD void elbo_forward_sequence(const ElboFwdArgs<R>& g, int b, ElboScratch<R, N, P>& s) {
    const long B = g.B;
    clear_scratch<ElboScratch<R, N, P>, R>(s);
    int st = elbo_prepare<R, N, P>(g, s);
    const R hlQ = half_logdet<R, N>(s.LQ), hlR = half_logdet<R, P>(s.LRm),
            hl0 = half_logdet<R, N>(s.L0);
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        if (!elbo_sample<R, N, P>(g, tb, s)) st |= ST_NONPD_SMOOTHED;
        if (g.z) gstore<R, N>(g.z + tb * N, s.z);
        // observation
        gload_mat<R, P, N>(g.Cs + tb * P * N, s.C);
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
            gload_mat<R, N, N>(g.As + tb * N * N, s.A);
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
    KW_FOR(e, D * D) {
        const int q = pad_index<D>(e);
        G[q] = R(0.5) * (u[e / D] * u[e % D] - Si[q]);
    }
    KW_SYNC();
    acc.template axpy<D>(w, G);
}

template <typename R, int N, int P>
KV_HD void elbo_backward_sequence(const ElboBwdArgs<R>& g, int b, ElboBwdScratch<R, N, P>& s,
                                  ElboGradAcc<R, N, P>& acc) {
    constexpr int NN = msize(N, N), PN = msize(P, N);
    const ElboFwdArgs<R>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    clear_scratch<ElboBwdScratch<R, N, P>, R>(s);
    elbo_prepare<R, N, P>(f, s);
    gemm<R, N, N, N, true, false, 0>(s.LiQ, s.LiQ, s.SiQ);
    gemm<R, P, P, P, true, false, 0>(s.LiR, s.LiR, s.SiR);
    gemm<R, N, N, N, true, false, 0>(s.Li0, s.Li0, s.Si0);
    gstore_zero<R, N * N>(g.d_As + ((long)T * B + b) * N * N);
    gstore_zero<R, P * N>(g.d_Cs + ((long)T * B + b) * P * N);
    vzero<R, N>(s.zn_back);  // w_{t+1} A_{t+1}^T u_{t+1}, zero for t = T-1
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        R wo, wt, wp;
        elbo_cotangents<R>(g, tb, wo, wt, wp);
        elbo_sample<R, N, P>(f, tb, s);  // s.z, s.L, s.e
        vcopy<R, N>(s.zn_back, s.zb);
        // observation term
        gload_mat<R, P, N>(f.Cs + tb * P * N, s.C);
        gload<R, P>(f.a + tb * P, s.ro);
        gemv<R, P, N, false, 2>(s.C, s.z, s.ro);
        whiten<R, P>(s.LiR, s.ro, s.y);  // s.ro = R^-1 r
        KW_FOR(i, P) g.d_a[tb * P + i] = -wo * s.ro[i];
        vzero<R, PN>(s.Cb);
        outer_add<R, P, N>(wo, s.ro, s.z, s.Cb);
        gstore_mat<R, P, N>(g.d_Cs + tb * P * N, s.Cb);
        gemv<R, N, P, true, 0>(s.C, s.ro, s.y);  // C^T u
        axpby<R, N>(R(1), s.zb, wo, s.y, s.zb);
        cov_grad<R, P>(wo, s.ro, s.SiR, s.G2, acc.dR);
        // transition term of this pair
        vzero<R, NN>(s.Ab);
        if (t == 0) {
            axpby<R, N>(R(1), s.z, R(-1), s.m0, s.r);
            whiten<R, N>(s.Li0, s.r, s.y);
            acc.dm0.template axpy<1>(wt, s.r);
            cov_grad<R, N>(wt, s.r, s.Si0, s.G1, acc.dP0);
        } else {
            // previous sample z_{t-1}: recompute into s.zp without disturbing s.L / s.e
            {
                const long tp = tb - B;
                gload_mat<R, N, N>(f.Ps + tp * N * N, s.T1);
                gload<R, N>(f.ms + tp * N, s.zp);
                gload<R, N>(f.eps + tp * N, s.y);
                cholesky<R, N>(s.T1, s.X);
                gemv<R, N, N, false, 1>(s.X, s.y, s.zp);
            }
            gload_mat<R, N, N>(f.As + tb * N * N, s.A);
            vcopy<R, N>(s.z, s.r);
            gemv<R, N, N, false, 2>(s.A, s.zp, s.r);
            whiten<R, N>(s.LiQ, s.r, s.y);  // s.r = Q^-1 r
            outer_add<R, N, N>(wt, s.r, s.zp, s.Ab);
            cov_grad<R, N>(wt, s.r, s.SiQ, s.G1, acc.dQ);
            // contribution to z_{t-1}: w_t A_t^T u_t, consumed by the next iteration
            gemv<R, N, N, true, 0>(s.A, s.r, s.zn_back);
            vscale<R, N>(wt, s.zn_back);
        }
        axpby<R, N>(R(1), s.zb, -wt, s.r, s.zb);
        gstore_mat<R, N, N>(g.d_As + tb * N * N, s.Ab);
        gstore<R, N>(g.d_ms + tb * N, s.zb);
        // z = ms + L eps; posterior term: -wp sum log diag L
        vzero<R, NN>(s.Lb);
        outer_add<R, N, N>(R(1), s.zb, s.e, s.Lb);
        KW_FOR(i, N) s.Lb[i * ld(N) + i] -= wp / s.L[i * ld(N) + i];
        KW_SYNC();
        tri_lower_inverse<R, N>(s.L, s.Li);
        vzero<R, NN>(s.Psb);
        cholesky_adjoint_add<R, N>(s.L, s.Li, s.Lb, s.X, s.Y, s.Psb);
        gstore_mat<R, N, N>(g.d_Ps + tb * N * N, s.Psb);
    }
}

// ---------------------------------------------------------------------------
// Mixing adjoint (second pass of the filter backward of this variant).
// For every row r = (t, b) of the (T+1)*B rows:
//   d_alpha[r][k] = <bar_As[r], A^k> + <bar_Cs[r], C^k>
//   d_A_K[k] += alpha[r][k] * bar_As[r],   d_C_K[k] += alpha[r][k] * bar_Cs[r]
// The device kernels are in wave_kernels.hip; this single-thread form defines
// the arithmetic and serves the host build.
// ---------------------------------------------------------------------------
template <typename R, int N, int P, int K>
inline void mix_adjoint_host(long rows, const R* alpha, const R* bar_As, const R* bar_Cs,
                             const R* A_K, const R* C_K, R* d_alpha, R* d_A_K, R* d_C_K) {
    constexpr int LA = N * N, LC = P * N;
    for (int i = 0; i < K * LA; ++i) d_A_K[i] = R(0);
    for (int i = 0; i < K * LC; ++i) d_C_K[i] = R(0);
    for (long r = 0; r < rows; ++r) {
        for (int k = 0; k < K; ++k) {
            R acc = R(0);
            for (int e = 0; e < LA; ++e) acc += bar_As[r * LA + e] * A_K[k * LA + e];
            for (int e = 0; e < LC; ++e) acc += bar_Cs[r * LC + e] * C_K[k * LC + e];
            d_alpha[r * K + k] = acc;
            const R w = alpha[r * K + k];
            for (int e = 0; e < LA; ++e) d_A_K[k * LA + e] += w * bar_As[r * LA + e];
            for (int e = 0; e < LC; ++e) d_C_K[k * LC + e] += w * bar_Cs[r * LC + e];
        }
    }
}

}  // namespace wave
}  // namespace kvae
