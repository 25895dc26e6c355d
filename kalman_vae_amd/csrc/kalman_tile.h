// Register-tile variant of the large-state (8 < z_dim <= 32) Kalman filter /
// RTS smoother / ELBO kernels for gfx950, fp32.
//
// Execution model: ONE WAVEFRONT OWNS ONE SEQUENCE and every matrix of the
// sequence lives in REGISTERS, in the accumulator layout of
// v_mfma_f32_32x32x2_f32: a (32 x 32) tile is 16 VGPRs, lane l = (j, h) with
// j = l & 31 and h = l >> 5 holds column j, rows rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h
// in register r.  Smaller matrices are zero-padded to the tile.
//
// The one product the matrix core gives without moving data is
//     tn(X, Y) = X^T Y
// (register r of X is the A operand, register r of Y the B operand of the
// r-th MFMA: both lanes' halves supply rows rho(r, 0) and rho(r, 1), i.e. the
// contraction runs over the ROW index of both tiles, and the result arrives in
// the same layout).  Everything is arranged around that:
//   * symmetric matrices (all covariances) are their own transpose;
//   * congruences  M^T S M = tn(tn(S, M), M)  need no transpose at all;
//   * inputs that are needed in both orientations are loaded twice from
//     global memory (second read hits L2): `load` gives the tile of X,
//     `load_t` the tile of X^T;
//   * a tile that was computed here and is needed transposed goes through a
//     4.2 KB LDS buffer (`transpose`), 32 LDS instructions.
// Vectors are kept in "lane form" (lane (j, h) holds x_j, both halves the same
// value); a matrix-vector product  X^T x  reads x_rho through v_readlane
// ("row form") and adds the two halves.
// Cholesky factor and triangular inverse run on a row-per-lane copy of the
// tile (32 registers) with v_readlane broadcasts, as in the LDS variant.
//
// No LDS beyond the transpose buffer, so residency is bounded by registers
// (up to 512 per lane at one wavefront per SIMD), not by the 160 KB LDS: the
// LDS variant (kalman_wave.h) keeps 60-97 KB of matrices per wavefront and
// runs one or two wavefronts per CU.
//
// Algorithm sources: kvae/state_space_model.py:284-352, :390-477 and
// kvae/kalman_vae.py:191-242 of the reference; adjoints hand-derived (same
// equations as kalman_core.h / kalman_wave.h).
#pragma once

#include <hip/hip_runtime.h>

#include "kalman_core.h"

namespace kvae {
namespace tile {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Tile {
    f32x16 v;
};
// x_{rho(r, h)} for r = 0..15: a vector in the layout of a tile column
struct Rows {
    float v[16];
};

#define KT_DEV __device__ __forceinline__
#define KT_UNROLL _Pragma("unroll")

constexpr int rho0(int r) { return (r & 3) + 8 * (r >> 2); }  // row of register r, half 0

KT_DEV int lane_col() { return threadIdx.x & 31; }
KT_DEV int lane_half() { return (threadIdx.x >> 5) & 1; }

// order the wavefront's LDS accesses for the compiler; a single wavefront
// executes them in program order, so no s_barrier is needed
KT_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

KT_DEV float bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
KT_DEV float other_half(float v) { return __shfl_xor(v, 32, 64); }
// sum of a lane-form vector's 32 entries (every lane gets the sum)
template <int CTRL>
KT_DEV float row_exchange(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
KT_DEV float hsum(float v) {
    // within each row of 16 lanes by DPP (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
    // row_mirror), then the neighbouring row of the same half through the LDS crossbar
    v += row_exchange<0xB1>(v);
    v += row_exchange<0x4E>(v);
    v += row_exchange<0x141>(v);
    v += row_exchange<0x140>(v);
    return v + __shfl_xor(v, 16, 64);
}

KT_DEV Tile tzero() {
    Tile t;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) t.v[r] = 0.f;
    return t;
}

// C {=, +=, -=} X^T Y, contraction over the first KD rows of both tiles
template <int KD, int MODE>
KT_DEV void tn(const Tile& X, const Tile& Y, Tile& C) {
    static_assert(KD % 8 == 0 && KD <= 32, "rows come in groups of 8");
    f32x16 acc;
    if (MODE == 0) {
        KT_UNROLL
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    } else {
        acc = C.v;
    }
    KT_UNROLL
    for (int r = 0; r < KD / 2; ++r)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(MODE == 2 ? -X.v[r] : X.v[r], Y.v[r], acc, 0, 0, 0);
    C.v = acc;
}

// ---- global memory <-> tiles (dense row-major ROWS x COLS matrices) -----------
// tile of X: v[r] = X[rho][j]
template <int ROWS, int COLS>
KT_DEV Tile load(const float* __restrict__ p) {
    static_assert(ROWS % 8 == 0 && COLS % 8 == 0 && ROWS <= 32 && COLS <= 32, "tile shapes");
    const int j = lane_col(), h = lane_half();
    Tile t;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) {
        float val = 0.f;
        if (rho0(r) < ROWS && (COLS == 32 || j < COLS)) val = p[(rho0(r) + 4 * h) * COLS + j];
        t.v[r] = val;
    }
    return t;
}
// tile of X^T: v[r] = X[j][rho]  (lane j reads row j, 16 bytes at a time)
template <int ROWS, int COLS>
KT_DEV Tile load_t(const float* __restrict__ p) {
    static_assert(ROWS % 8 == 0 && COLS % 8 == 0 && ROWS <= 32 && COLS <= 32, "tile shapes");
    const int j = lane_col(), h = lane_half();
    Tile t;
    KT_UNROLL
    for (int q = 0; q < 4; ++q) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (8 * q < COLS && (ROWS == 32 || j < ROWS))
            val = *reinterpret_cast<const float4*>(p + j * COLS + 8 * q + 4 * h);
        t.v[4 * q] = val.x;
        t.v[4 * q + 1] = val.y;
        t.v[4 * q + 2] = val.z;
        t.v[4 * q + 3] = val.w;
    }
    return t;
}
template <int ROWS, int COLS>
KT_DEV void store(float* __restrict__ p, const Tile& t) {
    const int j = lane_col(), h = lane_half();
    KT_UNROLL
    for (int r = 0; r < 16; ++r)
        if (rho0(r) < ROWS && (COLS == 32 || j < COLS)) p[(rho0(r) + 4 * h) * COLS + j] = t.v[r];
}
// the tile holds X^T; writes X (ROWS x COLS)
template <int ROWS, int COLS>
KT_DEV void store_t(float* __restrict__ p, const Tile& t) {
    const int j = lane_col(), h = lane_half();
    KT_UNROLL
    for (int q = 0; q < 4; ++q)
        if (8 * q < COLS && (ROWS == 32 || j < ROWS))
            *reinterpret_cast<float4*>(p + j * COLS + 8 * q + 4 * h) =
                make_float4(t.v[4 * q], t.v[4 * q + 1], t.v[4 * q + 2], t.v[4 * q + 3]);
}

template <int LEN>
KT_DEV float vload(const float* __restrict__ p) {
    const int j = lane_col();
    return (LEN == 32 || j < LEN) ? p[j] : 0.f;
}
template <int LEN>
KT_DEV void vstore(float* __restrict__ p, float x) {
    if (threadIdx.x < LEN) p[threadIdx.x] = x;
}

// ---- vectors --------------------------------------------------------------------
template <int LEN>
KT_DEV Rows rows_of(float x) {
    const int h = lane_half();
    Rows o;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) {
        if (rho0(r) < LEN) {
            const float a = bcast(x, rho0(r)), b = bcast(x, rho0(r) + 4);
            o.v[r] = h ? b : a;
        } else {
            o.v[r] = 0.f;
        }
    }
    return o;
}
// X^T x over the first KD rows of X (lane form result)
template <int KD>
KT_DEV float tv(const Tile& X, const Rows& x) {
    float acc = 0.f;
    KT_UNROLL
    for (int r = 0; r < KD / 2; ++r) acc += X.v[r] * x.v[r];
    return acc + other_half(acc);
}
template <int KD>
KT_DEV float tv(const Tile& X, float x) {
    return tv<KD>(X, rows_of<KD>(x));
}
// C += s * (u x^T - S) / 2 style updates are written inline by the callers;
// the plain outer product u x^T:
KT_DEV Tile outer(float s, const Rows& u, float x) {
    Tile t;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) t.v[r] = s * u.v[r] * x;
    return t;
}

// X^T through LDS (32 x 33 floats)
KT_DEV Tile transpose(const Tile& X, float* lds) {
    const int j = lane_col(), h = lane_half();
    wave_sync();
    KT_UNROLL
    for (int r = 0; r < 16; ++r) lds[(rho0(r) + 4 * h) * 33 + j] = X.v[r];
    wave_sync();
    Tile t;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) t.v[r] = lds[j * 33 + rho0(r) + 4 * h];
    wave_sync();
    return t;
}
KT_DEV void symmetrize(Tile& X, float* lds) {
    const Tile t = transpose(X, lds);
    KT_UNROLL
    for (int r = 0; r < 16; ++r) X.v[r] = (X.v[r] + t.v[r]) * 0.5f;
}

// ---- Cholesky factor and its inverse -------------------------------------------------
// S symmetric (N x N in the tile).  Outputs: LT = tile of L^T (v[r] = L[j][rho]),
// Li = tile of L^-1 (v[r] = Li[rho][j]) when INV, diag = L[j][j] in lane form
// (1 for j >= N).  Padding stays zero.  Returns false on a pivot <= 0.
template <int N, bool INV>
KT_DEV bool chol(const Tile& S, Tile& LT, Tile& Li, float& diag) {
    static_assert(N % 8 == 0 && N <= 32, "tile shapes");
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int i = lane_col(), h = lane_half();
    // row i of S, then of L, as register pairs: the dot products below run on
    // v_pk_fma_f32 with a pair of broadcast values as the scalar operand
    f2 row[N / 2];
    KT_UNROLL
    for (int r = 0; r < N / 2; ++r) {
        const float mine = S.v[r], theirs = other_half(S.v[r]);
        const int c0 = rho0(r), c1 = rho0(r) + 4;
        row[c0 / 2][c0 % 2] = h ? theirs : mine;
        row[c1 / 2][c1 % 2] = h ? mine : theirs;
    }
    float dmin = 1.f;  // smallest pivot seen (factorisation fails when <= 0)
    diag = 1.f;
    float rdiag = 1.f;
    KT_UNROLL
    for (int j = 0; j < N; ++j) {
        f2 acc = {0.f, 0.f};
        KT_UNROLL
        for (int q = 0; q < j / 2; ++q) {
            const f2 lj = {bcast(row[q][0], j), bcast(row[q][1], j)};
            acc += row[q] * lj;
        }
        float s = row[j / 2][j % 2] - (acc[0] + acc[1]);
        if (j % 2) s -= row[j / 2][0] * bcast(row[j / 2][0], j);
        float d = bcast(s, j);
        dmin = fminf(dmin, d);
        d = d > 0.f ? d : 1.f;
        const float inv = __builtin_amdgcn_rsqf(d);  // 1 ulp
        const float ljj = d * inv;
        // the lane comparisons are recomputed per column: hoisted out of the
        // caller's loop over t they would occupy 128 scalar registers
        int ii = i;
        asm volatile("" : "+v"(ii));
        row[j / 2][j % 2] = (ii == j) ? ljj : (ii > j ? s * inv : 0.f);
        diag = (ii == j) ? ljj : diag;
        rdiag = (ii == j) ? inv : rdiag;
    }
    const bool ok = dmin > 0.f;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int c0 = rho0(r) % N, c1 = (rho0(r) + 4) % N;
        LT.v[r] = rho0(r) < N ? (h ? row[c1 / 2][c1 % 2] : row[c0 / 2][c0 % 2]) : 0.f;
    }
    if (INV) {
        // the broadcasts of L below repeat those of the factorisation; keeping
        // all of them alive in scalar registers would spill, so hide the reuse
        KT_UNROLL
        for (int k = 0; k < N / 2; ++k) asm volatile("" : "+v"(row[k]));
        f2 x[N / 2];  // column i of L^-1
        KT_UNROLL
        for (int r = 0; r < N; ++r) {
            int ii = i;
            asm volatile("" : "+v"(ii));
            f2 acc = {0.f, 0.f};
            KT_UNROLL
            for (int q = 0; q < r / 2; ++q) {
                const f2 lr = {bcast(row[q][0], r), bcast(row[q][1], r)};
                acc += lr * x[q];
            }
            float v = ((ii == r) ? 1.f : 0.f) - (acc[0] + acc[1]);
            if (r % 2) v -= bcast(row[r / 2][0], r) * x[r / 2][0];
            x[r / 2][r % 2] = v * bcast(rdiag, r);
        }
        KT_UNROLL
        for (int r = 0; r < 16; ++r) {
            const int c0 = rho0(r) % N, c1 = (rho0(r) + 4) % N;
            Li.v[r] = rho0(r) < N ? (h ? x[c1 / 2][c1 % 2] : x[c0 / 2][c0 % 2]) : 0.f;
        }
    }
    return ok;
}

// ---------------------------------------------------------------------------
// ELBO terms (kvae/kalman_vae.py:191-242), sequential in t per sequence.
// ---------------------------------------------------------------------------

// Li^T tile of a constant covariance and sum_i log L_ii
template <int D>
KT_DEV bool whitener(const float* __restrict__ Sigma, float* lds, Tile& LiT, float& half_logdet) {
    const Tile S = load<D, D>(Sigma);
    Tile LT, Li;
    float dg;
    const bool ok = chol<D, true>(S, LT, Li, dg);
    LiT = transpose(Li, lds);
    half_logdet = hsum(logf(dg));
    return ok;
}

template <int N, int P>
__device__ void elbo_forward_sequence(const ElboFwdArgs<float>& g, int b, float* lds, int t0, int t1) {
    // steps t0 .. t1 - 1 of sequence b.  The terms of different steps depend on each other
    // only through the sample of the step before (the transition term), so a long sequence
    // is cut into pieces that run as wavefronts of their own: a sequence per wavefront leaves
    // two of the three wavefronts a SIMD has room for unused at 1024 sequences.
    const long B = g.B;
    int st = 0;
    Tile LiTQ, LiTR, LiT0;
    float hlQ, hlR, hl0;
    if (!whitener<N>(g.Q, lds, LiTQ, hlQ)) st |= ST_NONPD_NOISE;
    if (!whitener<P>(g.Rm, lds, LiTR, hlR)) st |= ST_NONPD_NOISE;
    if (!whitener<N>(g.P0, lds, LiT0, hl0)) st |= ST_NONPD_NOISE;
    const float m0 = vload<N>(g.m0);
    float zp = 0.f;
    const float c_obs = 0.5f * (float)P * log_2pi<float>(), c_lat = 0.5f * (float)N * log_2pi<float>();
    // the covariance the step starts with is requested one step ahead: its
    // factorisation is the first thing every step waits for
    if (t0 > 0) {
        // the sample of the step before this piece (the piece before computes it as well)
        const long tb = (long)(t0 - 1) * B + b;
        const Tile S = load<N, N>(g.Ps + tb * N * N);
        Tile LT, unused;
        float dg;
        chol<N, false>(S, LT, unused, dg);          // flagged by the piece that owns the step
        zp = vload<N>(g.ms + tb * N) + tv<N>(LT, vload<N>(g.eps + tb * N));
    }
    Tile S_next = load<N, N>(g.Ps + ((long)t0 * B + b) * N * N);
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        const long tb = (long)t * B + b;
        const Tile S = S_next;
        if (t + 1 < t1) S_next = load<N, N>(g.Ps + (tb + B) * N * N);
        const Tile CT = load_t<P, N>(g.Cs + tb * P * N);
        const float ms = vload<N>(g.ms + tb * N), e = vload<N>(g.eps + tb * N);
        const float a = vload<P>(g.a + tb * P);
        Tile LT, unused;
        float dg;
        if (!chol<N, false>(S, LT, unused, dg)) st |= ST_NONPD_SMOOTHED;
        const float z = ms + tv<N>(LT, e);
        if (g.z) vstore<N>(g.z + tb * N, z);
        const float ro = a - tv<N>(CT, z);
        const float yo = tv<P>(LiTR, ro);
        const float qo = hsum(yo * yo);
        float yt, hlt;
        if (t == 0) {
            yt = tv<N>(LiT0, z - m0);
            hlt = hl0;
        } else {
            const Tile AT = load_t<N, N>(g.As + tb * N * N);
            yt = tv<N>(LiTQ, z - tv<N>(AT, zp));
            hlt = hlQ;
        }
        const float qt = hsum(yt * yt), qp = hsum(e * e), hlp = hsum(logf(dg));
        if (threadIdx.x == 0) {
            g.obs[tb] = -(c_obs + 0.5f * qo) - hlR;
            g.trans[tb] = -(c_lat + 0.5f * qt) - hlt;
            g.post[tb] = -(c_lat + 0.5f * qp) - hlp;
        }
        zp = z;
    }
    if (st && g.status && threadIdx.x == 0) atomicOr(g.status, st);
}

// Sigma^-1 tile of a constant covariance
template <int D>
KT_DEV Tile precision(const float* __restrict__ Sigma) {
    const Tile S = load<D, D>(Sigma);
    Tile LT, Li, Si;
    float dg;
    chol<D, true>(S, LT, Li, dg);
    tn<D, 0>(Li, Li, Si);
    return Si;
}

// A tile kept in LDS (1024 floats; lane l owns slots 64 r + l): accumulators and constants
// that would otherwise hold 16 registers each for the whole sequence.
KT_DEV void lds_put(float* at, const Tile& X) {
    KT_UNROLL
    for (int r = 0; r < 16; ++r) at[64 * r + threadIdx.x] = X.v[r];
}
KT_DEV Tile lds_get(const float* at) {
    Tile X;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) X.v[r] = at[64 * r + threadIdx.x];
    return X;
}

// acc (in LDS) += w/2 (u u^T - Si)
KT_DEV void cov_grad(float w, const Rows& ur, float u, const Tile& Si, float* acc) {
    KT_UNROLL
    for (int r = 0; r < 16; ++r) acc[64 * r + threadIdx.x] += 0.5f * w * (ur.v[r] * u - Si.v[r]);
}

// Steps t0 .. t1 - 1 of sequence b, piece number `piece` (its parameter-gradient sums go to
// row piece * B + b of g.partial).  The steps depend on each other only through the sample of
// the step before and the cotangent the transition term of the step after sends back, so a
// long sequence is cut into pieces that run as wavefronts of their own; a piece factors one
// covariance more than it owns at either end.  acc: 3 tiles of LDS (dQ, dR, dP0).
template <int N, int P>
__device__ void elbo_backward_sequence(const ElboBwdArgs<float>& g, int b, float* lds, float* acc,
                                       int t0, int t1, int piece) {
    using PG = ElboParamGrads<N, P>;
    const ElboFwdArgs<float>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const int j = lane_col(), h = lane_half();
    float* const dQ = acc, * const dR = acc + 1024, * const dP0 = acc + 2048;
    const Tile SiQ = precision<N>(f.Q), SiR = precision<P>(f.Rm);
    lds_put(dQ, tzero());
    lds_put(dR, tzero());
    lds_put(dP0, tzero());
    const float m0 = vload<N>(f.m0);
    float dm0 = 0.f;
    float zn_back = 0.f;  // w_{t+1} A_{t+1}^T u_{t+1}
    float z_top = 0.f;    // the sample of step t1, which the piece after this one owns
    if (t1 == T) {
        store<N, N>(g.d_As + ((long)T * B + b) * N * N, tzero());
        store<P, N>(g.d_Cs + ((long)T * B + b) * P * N, tzero());
    } else {
        const long tb = (long)t1 * B + b;
        const Tile S = load<N, N>(f.Ps + tb * N * N);
        Tile LTn, unused;
        float dgn;
        chol<N, false>(S, LTn, unused, dgn);
        z_top = vload<N>(f.ms + tb * N) + tv<N>(LTn, vload<N>(f.eps + tb * N));
    }
    // factor, sample and noise of the pair being processed (cur) and of the
    // pair before it (prev), which the transition term needs: one factorisation
    // per iteration, at the top of the loop, for pair t - 1.
    Tile LT = tzero(), Li = tzero();
    float z = 0.f, e = 0.f;
    Tile S_next = load<N, N>(f.Ps + ((long)(t1 - 1) * B + b) * N * N);   // one step ahead
#pragma unroll 1
    for (int t = t1; t >= t0; --t) {
        Tile LTp = tzero(), Lip = tzero();
        float zp = 0.f, ep = 0.f, dgp = 1.f;
        if (t > 0) {
            const long tp = (long)(t - 1) * B + b;
            const Tile S = S_next;
            if (t > 1) S_next = load<N, N>(f.Ps + (tp - B) * N * N);
            const float ms = vload<N>(f.ms + tp * N);
            ep = vload<N>(f.eps + tp * N);
            chol<N, true>(S, LTp, Lip, dgp);
            zp = ms + tv<N>(LTp, ep);
        }
        if (t == t1 && t1 < T) {
            // what the transition term of step t1 sends back to the sample of step t1 - 1
            const long tb = (long)t1 * B + b;
            float wo, wt, wp;
            elbo_cotangents<float>(g, tb, wo, wt, wp);
            const Tile AT = load_t<N, N>(f.As + tb * N * N);
            const Tile A = load<N, N>(f.As + tb * N * N);
            const float ut = tv<N>(SiQ, z_top - tv<N>(AT, zp));
            zn_back = wt * tv<N>(A, rows_of<N>(ut));
        }
        if (t < t1) {
            const long tb = (long)t * B + b;
            float wo, wt, wp;
            elbo_cotangents<float>(g, tb, wo, wt, wp);
            float zb = zn_back;
            // observation term
            {
                const Tile CT = load_t<P, N>(f.Cs + tb * P * N);
                const Tile C = load<P, N>(f.Cs + tb * P * N);
                const float ro = vload<P>(f.a + tb * P) - tv<N>(CT, z);
                const float u = tv<P>(SiR, ro);
                if (threadIdx.x < P) g.d_a[tb * P + threadIdx.x] = -wo * u;
                const Rows ur = rows_of<P>(u);
                store<P, N>(g.d_Cs + tb * P * N, outer(wo, ur, z));
                zb += wo * tv<P>(C, ur);
                cov_grad(wo, ur, u, SiR, dR);
            }
            // transition term
            float ut;
            if (t == 0) {
                const Tile S0 = precision<N>(f.P0);
                ut = tv<N>(S0, z - m0);
                const Rows utr = rows_of<N>(ut);
                dm0 += wt * ut;
                cov_grad(wt, utr, ut, S0, dP0);
                store<N, N>(g.d_As + tb * N * N, tzero());
                zn_back = 0.f;
            } else {
                const Tile AT = load_t<N, N>(f.As + tb * N * N);
                const Tile A = load<N, N>(f.As + tb * N * N);
                ut = tv<N>(SiQ, z - tv<N>(AT, zp));
                const Rows utr = rows_of<N>(ut);
                store<N, N>(g.d_As + tb * N * N, outer(wt, utr, zp));
                cov_grad(wt, utr, ut, SiQ, dQ);
                zn_back = wt * tv<N>(A, utr);
            }
            zb -= wt * ut;
            vstore<N>(g.d_ms + tb * N, zb);
            // z = ms + L eps and the posterior term -wp sum log L_ii, through the
            // factorisation:  d_Ps = L^-T Psi L^-1,  Psi = sym(Phi(L^T tril(Lbar))),
            // Lbar = zb eps^T - diag(wp / L_ii)  =>  Phi[i][j] = g_i eps_j (i > j),
            // (g_i eps_i - wp)/2 (i = j) with g = L^T zb.
            const Tile L = transpose(LT, lds);
            const float gv = tv<N>(L, zb);
            const Rows gr = rows_of<N>(gv), er = rows_of<N>(e);
            Tile Psi;
            KT_UNROLL
            for (int r = 0; r < 16; ++r) {
                const int row = rho0(r) + 4 * h;
                const float lo = 0.5f * gr.v[r] * e, up = 0.5f * gv * er.v[r];
                const float dd = (N == 32 || j < N) ? 0.5f * (gv * e - wp) : 0.f;
                Psi.v[r] = row > j ? lo : (row < j ? up : dd);
            }
            Tile W, dPs;
            tn<N, 0>(Psi, Li, W);
            tn<N, 0>(W, Li, dPs);
            store<N, N>(g.d_Ps + tb * N * N, dPs);
        }
        LT = LTp;
        Li = Lip;
        z = zp;
        e = ep;
    }
    float* row = g.partial + ((long)piece * B + b) * PG::LEN;
    store<N, N>(row + PG::OFF_Q, lds_get(dQ));
    store<P, P>(row + PG::OFF_R, lds_get(dR));
    vstore<N>(row + PG::OFF_M0, dm0);
    store<N, N>(row + PG::OFF_P0, lds_get(dP0));
}


KT_DEV void tadd(Tile& X, const Tile& Y) {
    KT_UNROLL
    for (int r = 0; r < 16; ++r) X.v[r] += Y.v[r];
}

// Adjoint of  z = ms + chol(S) e  w.r.t. S for the cotangent zb of z, plus the
// adjoint of  -wp sum_i log L_ii:  returns  L^-T Psi L^-1  with
// Psi = sym(Phi(L^T tril(zb e^T - diag(wp / L_ii)))), i.e. Phi[i][j] = g_i e_j
// for i > j and (g_i e_i - wp)/2 on the diagonal, g = L^T zb.  LT, Li from chol().
template <int N, int MODE>
KT_DEV void chol_sample_adjoint(const Tile& LT, const Tile& Li, float zb, float e, float wp,
                                float* lds, Tile& out) {
    const int j = lane_col(), h = lane_half();
    const Tile L = transpose(LT, lds);
    const float gv = tv<N>(L, zb);
    const Rows gr = rows_of<N>(gv), er = rows_of<N>(e);
    Tile Psi;
    KT_UNROLL
    for (int r = 0; r < 16; ++r) {
        const int row = rho0(r) + 4 * h;
        const float lo = 0.5f * gr.v[r] * e, up = 0.5f * gv * er.v[r];
        const float dd = (N == 32 || j < N) ? 0.5f * (gv * e - wp) : 0.f;
        Psi.v[r] = row > j ? lo : (row < j ? up : dd);
    }
    Tile W;
    tn<N, 0>(Psi, Li, W);
    tn<N, MODE>(W, Li, out);
}

// ---------------------------------------------------------------------------
// RTS smoother (kvae/state_space_model.py:390-477), t = T-1 .. 0.
// With J = Pf An^T Pn^-1:  W = An^T Pn^-1 = tn(An, Pi),  J^T = W^T Pf = tn(W, Pf),
// J d = tv(J^T, d),  J D J^T = tn(tn(D, J^T), J^T).
// ---------------------------------------------------------------------------
template <int N, int P>
__device__ void smooth_forward_sequence(const SmoothFwdArgs<float>& g, int b, float* lds) {
    const long B = g.B;
    const int T = g.T;
    int st = 0;
    Tile LRT = tzero();
    if (g.eps_o) LRT = load_t<P, P>(g.LR);
    float ms = 0.f;
    Tile Ps = tzero();
    // the predicted covariance whose factorisation opens a step, requested a step ahead
    Tile Pn_next = T > 1 ? load<N, N>(g.Pn + ((long)(T - 2) * B + b) * N * N) : tzero();
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        const bool last = t == T - 1;
        const Tile Pf = load<N, N>(g.Pf + tb * N * N);
        const float m = vload<N>(g.m + tb * N);
        const Tile CT = load_t<P, N>(g.Cs + (last ? tb + B : tb) * P * N);
        if (last) {
            ms = m;
            Ps = Pf;
        } else {
            const Tile Pn = Pn_next;
            if (t > 0) Pn_next = load<N, N>(g.Pn + (tb - B) * N * N);
            const Tile An = load<N, N>(g.As + (tb + B) * N * N);
            const float mn = vload<N>(g.mn + tb * N);
            Tile LT, Li, Pi, W, JT, X;
            float dg;
            if (!chol<N, true>(Pn, LT, Li, dg)) st |= ST_NONPD_PREDICTED;
            tn<N, 0>(Li, Li, Pi);
            tn<N, 0>(An, Pi, W);
            tn<N, 0>(W, Pf, JT);
            ms = m + tv<N>(JT, ms - mn);
            KT_UNROLL
            for (int r = 0; r < 16; ++r) Ps.v[r] -= Pn.v[r];
            tn<N, 0>(Ps, JT, X);
            Ps = Pf;
            tn<N, 1>(X, JT, Ps);
            if (g.symmetrize) symmetrize(Ps, lds);
        }
        vstore<N>(g.ms + tb * N, ms);
        store<N, N>(g.Ps + tb * N * N, Ps);
        float z = ms;
        if (g.eps_s) {
            Tile LT, unused;
            float dg;
            if (!chol<N, false>(Ps, LT, unused, dg)) st |= ST_NONPD_SMOOTHED;
            z += tv<N>(LT, vload<N>(g.eps_s + tb * N));
        }
        float ar = tv<N>(CT, z);
        if (g.eps_o) ar += tv<P>(LRT, vload<P>(g.eps_o + tb * P));
        vstore<N>(g.zs + tb * N, z);
        vstore<P>(g.ares + tb * P, ar);
    }
    if (st && g.status && threadIdx.x == 0) atomicOr(g.status, st);
}

// Adjoint, t = 0 .. T-1, carrying (msb, G) = cotangents of (ms_t, Ps_t).
template <int N, int P>
__device__ void smooth_backward_sequence(const SmoothBwdArgs<float>& g, int b, float* lds) {
    const SmoothFwdArgs<float>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const bool emit = g.g_ares || g.g_zs;
    Tile dLR = tzero();
    {
        const Tile zero = tzero();
        store<N, N>(g.d_As + (long)b * N * N, zero);
        store<N, N>(g.d_As + ((long)T * B + b) * N * N, zero);
        store<P, N>(g.d_Cs + ((long)(T - 1) * B + b) * P * N, zero);
        vstore<N>(g.d_mn + ((long)(T - 1) * B + b) * N, 0.f);
        store<N, N>(g.d_Pn + ((long)(T - 1) * B + b) * N * N, zero);
    }
    float msb = 0.f;
    Tile G = tzero();
    // the smoothed covariance of the next step is requested a step ahead: it is both the
    // first thing the next step factorises and this step's Ps_{t+1}
    Tile Ps_next = load<N, N>(f.Ps + (long)b * N * N);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const long tb = (long)t * B + b;
        const bool last = t == T - 1;
        const float ms = vload<N>(f.ms + tb * N);
        const Tile Ps = Ps_next;
        if (!last) Ps_next = load<N, N>(f.Ps + (tb + B) * N * N);
        // the predicted covariance is factorised right after the emission adjoint
        const Tile Pn = last ? tzero() : load<N, N>(f.Pn + tb * N * N);
        if (g.g_ms) msb += vload<N>(g.g_ms + tb * N);
        if (g.g_Ps) tadd(G, load<N, N>(g.g_Ps + tb * N * N));
        // emission adjoint: zs = ms + chol(Ps) eps_s, ares = C zs + LR eps_o
        float* dC = g.d_Cs + (last ? tb + B : tb) * P * N;
        if (emit) {
            const float* Cp = f.Cs + (last ? tb + B : tb) * P * N;
            float zb = g.g_zs ? vload<N>(g.g_zs + tb * N) : 0.f;
            Tile LT = tzero(), Li = tzero();
            float dg, e = 0.f;
            if (f.eps_s) {
                chol<N, true>(Ps, LT, Li, dg);
                e = vload<N>(f.eps_s + tb * N);
            }
            if (g.g_ares) {
                const float ab = vload<P>(g.g_ares + tb * P);
                const float z = ms + tv<N>(LT, e);
                const Rows abr = rows_of<P>(ab);
                store<P, N>(dC, outer(1.f, abr, z));
                zb += tv<P>(load<P, N>(Cp), abr);
                if (f.eps_o && g.partial) {
                    const float eo = vload<P>(f.eps_o + tb * P);
                    KT_UNROLL
                    for (int r = 0; r < 16; ++r) dLR.v[r] += abr.v[r] * eo;
                }
            } else {
                store<P, N>(dC, tzero());
            }
            msb += zb;
            if (f.eps_s) chol_sample_adjoint<N, 1>(LT, Li, zb, e, 0.f, lds, G);
        } else {
            store<P, N>(dC, tzero());
        }
        if (last) {
            vstore<N>(g.d_m + tb * N, msb);
            store<N, N>(g.d_P + tb * N * N, G);
        } else {
            if (t < g.burn_in) {
                msb = 0.f;
                G = tzero();
            }
            const Tile Pf = load<N, N>(f.Pf + tb * N * N);
            const Tile An = load<N, N>(f.As + (tb + B) * N * N);
            const Tile AnT = load_t<N, N>(f.As + (tb + B) * N * N);
            Tile D = Ps_next;
            const float d = vload<N>(f.ms + (tb + B) * N) - vload<N>(f.mn + tb * N);
            Tile LT, Li, Pi, W, WT, J;
            float dg;
            chol<N, true>(Pn, LT, Li, dg);
            tn<N, 0>(Li, Li, Pi);
            tn<N, 0>(An, Pi, W);
            WT = transpose(W, lds);
            tn<N, 0>(Pf, W, J);
            KT_UNROLL
            for (int r = 0; r < 16; ++r) D.v[r] -= Pn.v[r];
            // Gsum = G + G^T (and G made symmetric when the forward pass did so)
            Tile Gsum = transpose(G, lds);
            KT_UNROLL
            for (int r = 0; r < 16; ++r) Gsum.v[r] += G.v[r];
            if (f.symmetrize) {
                KT_UNROLL
                for (int r = 0; r < 16; ++r) G.v[r] = 0.5f * Gsum.v[r];
            }
            vstore<N>(g.d_m + tb * N, msb);
            // Jb = msb d^T + (G + G^T) J D and its transpose
            Tile MT, Jb, JbT;
            tn<N, 0>(J, Gsum, MT);
            const Rows msbr = rows_of<N>(msb);
            Jb = outer(1.f, msbr, d);
            tn<N, 1>(MT, D, Jb);
            JbT = transpose(Jb, lds);
            const float db = tv<N>(J, msb);
            Tile X1, Db;
            tn<N, 0>(G, J, X1);
            tn<N, 0>(X1, J, Db);
            // Pfb = G + Jb W^T
            tn<N, 1>(JbT, WT, G);
            store<N, N>(g.d_P + tb * N * N, G);
            Tile Wb, WbT, Anb, PibT, Z;
            tn<N, 0>(Pf, Jb, Wb);
            WbT = transpose(Wb, lds);
            tn<N, 0>(Pi, WbT, Anb);
            store<N, N>(g.d_As + (tb + B) * N * N, Anb);
            tn<N, 0>(Wb, AnT, PibT);
            tn<N, 0>(PibT, Pi, Z);
            Tile Pnb;
            KT_UNROLL
            for (int r = 0; r < 16; ++r) Pnb.v[r] = -Db.v[r];
            tn<N, 2>(Pi, Z, Pnb);
            store<N, N>(g.d_Pn + tb * N * N, Pnb);
            vstore<N>(g.d_mn + tb * N, -db);
            msb = db;
            G = Db;
        }
    }
    if (g.partial) store<P, P>(g.partial + (long)b * (P * P), dLR);
}


// ---------------------------------------------------------------------------
// Filter (kvae/state_space_model.py:284-352).
// With CT = tile of C^T and Pp symmetric:  F = Pp C^T = tn(Pp, CT),
// M = C Pp = tn(CT, Pp) = F^T,  S = R + C Pp C^T = R + tn(F, CT),
// G r = F Si r = tv(M, tv(Si, r)),  G M = M^T Si M = tn(tn(Si, M), M),
// An Pf An^T = tn(tn(Pf, AnT), AnT) with AnT = tile of An^T.
// ---------------------------------------------------------------------------

// tile of (sum_k w[k] X_k)^T for K dense row-major (ROWS x COLS) bases; w is
// wave-uniform (scalar loads), lane j reads row j of every basis.
template <int ROWS, int COLS>
KT_DEV Tile mix_t(int K, const float* __restrict__ w, const float* __restrict__ bases) {
    const int j = lane_col(), h = lane_half();
    Tile t = tzero();
    if (ROWS == 32 || j < ROWS) {
#pragma unroll 2
        for (int k = 0; k < K; ++k) {
            const float wk = w[k];
            const float* p = bases + (long)k * ROWS * COLS + j * COLS + 4 * h;
            KT_UNROLL
            for (int q = 0; q < 4; ++q) {
                if (8 * q < COLS) {
                    const float4 val = *reinterpret_cast<const float4*>(p + 8 * q);
                    t.v[4 * q] += wk * val.x;
                    t.v[4 * q + 1] += wk * val.y;
                    t.v[4 * q + 2] += wk * val.z;
                    t.v[4 * q + 3] += wk * val.w;
                }
            }
        }
    }
    return t;
}

// F, M, S^-1 and the innovation of one measurement update
template <int N, int P>
KT_DEV bool update_terms(const Tile& Pp, const Tile& CT, const Tile& Rm, float a, float mp,
                         float* lds, Tile& F, Tile& M, Tile& Si, float& r) {
    tn<N, 0>(Pp, CT, F);
    M = transpose(F, lds);  // a transpose through LDS costs a quarter of a second product
    Tile S = Rm;
    tn<N, 1>(F, CT, S);
    Tile LT, Li;
    float dg;
    const bool ok = chol<P, true>(S, LT, Li, dg);
    tn<P, 0>(Li, Li, Si);
    r = a - tv<N>(CT, mp);
    return ok;
}

template <int N, int P>
__device__ void filter_forward_sequence(const FilterFwdArgs<float>& g, int K, int b, float* lds) {
    const long B = g.B;
    const Tile Rm = load<P, P>(g.Rm), Q = load<N, N>(g.Q);
    float mp = vload<N>(g.m_in + (g.carry_per_sequence ? (long)b * N : 0));
    Tile Pp = load<N, N>(g.P_in + (g.carry_per_sequence ? (long)b * N * N : 0));
    if (g.As) store_t<N, N>(g.As + (long)b * N * N, mix_t<N, N>(K, g.alpha + (long)b * K, g.A_K));
    int st = 0;
#pragma unroll 1
    for (int t = 0; t < g.T; ++t) {
        const long tb = (long)t * B + b;
        const Tile CT = mix_t<P, N>(K, g.alpha + tb * K, g.C_K);
        const Tile AnT = mix_t<N, N>(K, g.alpha + (tb + B) * K, g.A_K);
        if (g.Cs) store_t<P, N>(g.Cs + tb * P * N, CT);
        if (g.As) store_t<N, N>(g.As + (tb + B) * N * N, AnT);
        const float a = vload<P>(g.a + tb * P);
        const float mu = g.mask ? g.mask[tb] : 1.f;
        Tile F, M, Si, X;
        float r;
        if (!update_terms<N, P>(Pp, CT, Rm, a, mp, lds, F, M, Si, r)) st |= ST_NONPD_INNOVATION;
        const float m = mp + mu * tv<P>(M, tv<P>(Si, r));
        tn<P, 0>(Si, M, X);
        Tile Y;
        tn<P, 0>(X, M, Y);
        Tile Pf;
        KT_UNROLL
        for (int q = 0; q < 16; ++q) Pf.v[q] = Pp.v[q] - mu * Y.v[q];
        if (g.symmetrize) symmetrize(Pf, lds);
        mp = tv<N>(AnT, m);
        tn<N, 0>(Pf, AnT, X);
        Pp = Q;
        tn<N, 1>(X, AnT, Pp);
        if (g.symmetrize) symmetrize(Pp, lds);
        vstore<N>(g.m + tb * N, m);
        store<N, N>(g.Pf + tb * N * N, Pf);
        vstore<N>(g.mn + tb * N, mp);
        store<N, N>(g.Pn + tb * N * N, Pp);
    }
    if (g.Cs)
        store_t<P, N>(g.Cs + ((long)g.T * B + b) * P * N,
                      mix_t<P, N>(K, g.alpha + ((long)g.T * B + b) * K, g.C_K));
    if (st && g.status && threadIdx.x == 0) atomicOr(g.status, st);
}

template <int ROWS, int COLS>
KT_DEV Tile load_or_zero(const float* __restrict__ base, long off) {
    return base ? load<ROWS, COLS>(base + off) : tzero();
}

// Adjoint, t = T-1 .. 0, carrying (mpb, Ppb) = cotangents of the predicted
// (mean, covariance) entering step t.  Needs the forward pass's stored mixed
// matrices (f.As, f.Cs); writes the gradients w.r.t. them (bar_As, bar_Cs) for
// the separate mixing-adjoint pass, and one row [dQ | dR] of partial sums.
template <int N, int P>
__device__ void filter_backward_sequence(const FilterBwdArgs<float>& g, int b, float* lds) {
    const FilterFwdArgs<float>& f = g.f;
    const long B = f.B;
    const int T = f.T;
    const Tile Rm = load<P, P>(f.Rm);
    Tile dQ = tzero(), dR = tzero();
    store<P, N>(g.bar_Cs + ((long)T * B + b) * P * N,
                load_or_zero<P, N>(g.g_Cs, ((long)T * B + b) * P * N));
    float mpb = 0.f;
    Tile Ppb = tzero();
    Tile An_next = load<N, N>(f.As + ((long)T * B + b) * N * N);   // a step ahead
#pragma unroll 1
    for (int t = T - 1; t >= 0; --t) {
        const long tb = (long)t * B + b;
        const Tile C = load<P, N>(f.Cs + tb * P * N), CT = load_t<P, N>(f.Cs + tb * P * N);
        const Tile An = An_next;
        if (t > 0) An_next = load<N, N>(f.As + tb * N * N);
        const float a = vload<P>(f.a + tb * P);
        float mp;
        Tile Pp;
        if (t > 0) {
            mp = vload<N>(f.mn + (tb - B) * N);
            Pp = load<N, N>(f.Pn + (tb - B) * N * N);
        } else {
            mp = vload<N>(f.m_in + (f.carry_per_sequence ? (long)b * N : 0));
            Pp = load<N, N>(f.P_in + (f.carry_per_sequence ? (long)b * N * N : 0));
        }
        const float m = vload<N>(f.m + tb * N);
        const Tile Pf = load<N, N>(f.Pf + tb * N * N);
        float mb = g.g_m ? vload<N>(g.g_m + tb * N) : 0.f;
        Tile Pb = load_or_zero<N, N>(g.g_P, tb * N * N);
        if (g.g_mn) mpb += vload<N>(g.g_mn + tb * N);
        if (g.g_Pn) tadd(Ppb, load<N, N>(g.g_Pn + tb * N * N));
        if (t < g.burn_in) {
            mb = 0.f;
            mpb = 0.f;
            Pb = tzero();
            Ppb = tzero();
        }
        // time update adjoint:  mn = An m,  Pn = An Pf An^T + Q
        {
            Tile Hs = transpose(Ppb, lds);  // Ppb + Ppb^T
            KT_UNROLL
            for (int r = 0; r < 16; ++r) Hs.v[r] += Ppb.v[r];
            if (f.symmetrize) {
                KT_UNROLL
                for (int r = 0; r < 16; ++r) Ppb.v[r] = 0.5f * Hs.v[r];
            }
            Tile Ab = load_or_zero<N, N>(g.g_As, (tb + B) * N * N);
            tadd(Ab, outer(1.f, rows_of<N>(mpb), m));
            Tile X;
            tn<N, 0>(An, Hs, X);
            tn<N, 1>(X, Pf, Ab);
            store<N, N>(g.bar_As + (tb + B) * N * N, Ab);
            mb += tv<N>(An, mpb);
            tn<N, 0>(Ppb, An, X);
            tn<N, 1>(X, An, Pb);
            tadd(dQ, Ppb);
        }
        // measurement update adjoint
        Tile F, M, Si, G;
        float r;
        update_terms<N, P>(Pp, CT, Rm, a, mp, lds, F, M, Si, r);
        tn<P, 0>(M, Si, G);
        const float mu = f.mask ? f.mask[tb] : 1.f;
        Tile PbT = transpose(Pb, lds);
        if (f.symmetrize) {
            KT_UNROLL
            for (int q = 0; q < 16; ++q) {
                Pb.v[q] = 0.5f * (Pb.v[q] + PbT.v[q]);
                PbT.v[q] = Pb.v[q];
            }
        }
        Tile Cb = load_or_zero<P, N>(g.g_Cs, tb * P * N);
        const float mob = mu * mb;
        mpb = mb;
        Ppb = Pb;
        Tile Pob, PobT;
        KT_UNROLL
        for (int q = 0; q < 16; ++q) {
            Pob.v[q] = mu * Pb.v[q];
            PobT.v[q] = mu * PbT.v[q];
        }
        const Rows mobr = rows_of<N>(mob);
        Tile Gb = outer(1.f, mobr, r);
        tn<N, 2>(PobT, F, Gb);
        const Tile GbT = transpose(Gb, lds);
        const float rb = tv<N>(G, mob);
        vstore<P>(g.d_a + tb * P, rb);
        const Rows rbr = rows_of<P>(rb);
        KT_UNROLL
        for (int q = 0; q < 16; ++q) Cb.v[q] -= rbr.v[q] * mp;
        mpb -= tv<P>(C, rbr);
        Tile Mb, MbT;
        tn<N, 0>(G, Pob, Mb);
        MbT = transpose(Mb, lds);
        tn<N, 2>(MbT, Pp, Cb);
        tn<P, 2>(C, Mb, Ppb);
        Tile Fb, FbT, Sib, U, Sb;
        tn<P, 0>(GbT, Si, Fb);
        FbT = transpose(Fb, lds);
        tn<N, 0>(F, Gb, Sib);
        tn<P, 0>(Sib, Si, U);
        Sb = tzero();
        tn<P, 2>(U, Si, Sb);
        tn<P, 1>(FbT, C, Ppb);
        tn<N, 1>(Fb, Pp, Cb);
        Tile Ssum = transpose(Sb, lds);
        tadd(Ssum, Sb);
        tn<P, 1>(Ssum, M, Cb);
        tn<P, 0>(Sb, C, U);
        tn<P, 1>(U, C, Ppb);
        tadd(dR, Sb);
        store<P, N>(g.bar_Cs + tb * P * N, Cb);
    }
    store<N, N>(g.bar_As + (long)b * N * N, load_or_zero<N, N>(g.g_As, (long)b * N * N));
    vstore<N>(g.d_m_in + (long)b * N, mpb);
    store<N, N>(g.d_P_in + (long)b * N * N, Ppb);
    float* row = g.partial + (long)b * (N * N + P * P);
    store<N, N>(row, dQ);
    store<P, P>(row + N * N, dR);
}

}  // namespace tile
}  // namespace kvae
