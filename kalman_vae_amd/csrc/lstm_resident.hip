// One LSTM layer (hidden size 128) over a whole sequence in ONE launch per direction,
// with the layer's recurrent matrix held in registers for all T steps.
//
// Reference: the recurrence of torch.nn.LSTM as used by the dynamics-parameter network
// (kvae/dynamics_parameter_network.py:46-58; gate order i, f, g, o).
//
// The 128 x 512 recurrent matrix is 256 KiB: too large for the 160 KiB LDS of a compute
// unit, but a compute unit has 512 KiB of vector registers.  A workgroup of four
// wavefronts (one per SIMD, 512 registers per lane each) owns 16 sequences; wavefront w
// owns hidden units [32w, 32w+32) and keeps its 128 x 128 slice of the matrix in 256
// registers per lane, laid out as the B operands of v_mfma_f32_16x16x4_f32 (exact fp32),
// so a step issues 256 matrix instructions whose operands are all registers plus one LDS
// read per four of them -- nothing is streamed from L2 inside the time loop.  h_t is
// exchanged between the wavefronts through LDS (double buffered, one barrier per step),
// c_t stays in registers.  A lane owns rows 4s..4s+3 (s = lane >> 4) and the adjacent
// hidden units ub, ub+1 (ub = 32w + 2 (lane & 15)): global accesses are 8 bytes wide.
//
// A layer's input projection is not part of the recurrence: for a narrow input (the first
// layer sees a_dim <= 4 features) it is formed in the kernel; otherwise the caller
// computes it for all steps with one library GEMM and passes it in `gates`.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/kvae_kalman.h"
#include "lstm_common.h"

namespace kvae {

constexpr int RH = 128;            // hidden size
constexpr int RG4 = 4 * RH;
constexpr int RLDA [[maybe_unused]] = RH + 2;   // LDS row stride of an h tile: rows 2 banks apart
constexpr int RLDG [[maybe_unused]] = RG4 + 2;   // LDS row stride of a gate-gradient tile

// Workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for every
// outstanding global store of the wavefront (its release fence covers global memory);
// here each step leaves ~50 KB of streaming stores in flight per workgroup that nothing
// in the workgroup reads back, and draining them at the barrier would cost more than the
// step's matrix instructions.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct GateRows {                  // one lane's 4 rows x 2 units of the four gates
    float2 v[4][4];                // [row][gate]
};

#ifdef KVAE_LSTM_BF16X3
// Split-precision build of this translation unit.  The fp32 matrix instruction
// (16x16x4, 32 cycles) is replaced by three bf16 ones (16x16x32, 16 cycles each, eight
// times the depth): with x = x_hi + x_lo, both bf16, x_lo = bf16(x - x_hi),
//     a b  ~  a_hi b_hi + a_lo b_hi + a_hi b_lo          (accumulated in fp32)
// drops a_lo b_lo and the two representation residuals, each below 2^-16 |a b|: a product
// is good to about 1e-5 relative (5e-5 at worst; fp32: 6e-8), a 128-term gate sum to about
// 1e-6 of the sum of its terms' magnitudes.  Operands use the same registers as the fp32 form (a hi and a
// lo bf16 per fp32 value); a step needs 96 matrix instructions of 16 cycles instead of 256
// of 32.  The kernels of this build carry `split` in their names.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
constexpr int RLDB = RH + 8;       // bf16 elements per LDS row of an h tile (272 B: rows 4 banks apart)
constexpr int RLDGB [[maybe_unused]] = RG4 + 8;   // ... of a gate-gradient tile (1040 B)

// (x, y) -> packed bf16 pairs hi = (bf16 x, bf16 y), lo = (bf16 (x - hi.x), bf16 (y - hi.y))
__device__ __forceinline__ void split_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const f32x2v v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2v r = {x - __builtin_bit_cast(float, hi << 16),
                      y - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
// eight fp32 values -> the hi and lo operand registers of one lane
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(x[2 * j], x[2 * j + 1], h[j], l[j]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
    hi = __builtin_bit_cast(bf16x8, hv);
    lo = __builtin_bit_cast(bf16x8, lv);
}
// a tile in LDS as its hi and lo halves
template <int ELEMS>
struct SplitTile {
    __bf16 hi[ELEMS], lo[ELEMS];
};
typedef SplitTile<16 * RLDB> HTile;
typedef SplitTile<16 * RLDGB> GTile;
#define KVAE_HBUF_T HTile (&hbuf)[2]
#define KVAE_GTILE_T GTile (&tile)[2]
#else
#define KVAE_HBUF_T float (&hbuf)[2][16 * RLDA]
#define KVAE_GTILE_T float (&tile)[2][16 * RLDG]
#endif

// grid = ceil(B/16), block = 256.  P > 0: the layer input x (T,B,P) with Wih (4H,P) and
// bias (4H) = b_ih + b_hh is projected in the kernel; P == 0: gates (T,B,4H) holds the
// input projection (bias included) on entry.  With store_gates the activated gates of
// every step are written to `gates` (in place for P == 0).
// FULL: all 16 rows of the workgroup's tile exist.  Then no access is predicated, the loop
// body is one basic block and the compiler counts outstanding memory operations exactly;
// with per-row predicates it waits for ALL of them (the previous step's ~50 KB of stores
// included) before it touches the next step's prefetched inputs.
template <int P, bool FULL, bool GATES>
__device__ __forceinline__ void
lstm128_forward_body(int T, int B, const float* __restrict__ x, const float* __restrict__ Wih,
                     const float* __restrict__ bias, const float* __restrict__ WhhT, float* gates,
                     float* __restrict__ h, float* __restrict__ c,
                     const float* __restrict__ h_init, const float* __restrict__ c_init,
                     KVAE_HBUF_T) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cc = lane & 15, s = lane >> 4;
#ifdef KVAE_LSTM_TILE_LOOP
    const int u0 = 32 * wave, ub = u0 + 2 * cc;
#else
    const int r0 = blockIdx.x * 16, u0 = 32 * wave, ub = u0 + 2 * cc;
#endif
    const long BH = (long)B * RH, BG = (long)B * RG4;

#ifdef KVAE_LSTM_BF16X3
    // this wavefront's slice of the recurrent matrix as B operands of the 16x16x32
    // instruction: w*[gate][hf][kk][j] multiplies h[., 32 kk + 8 s + j] into the gate
    // pre-activation of unit ub + hf
    bf16x8 whi[4][2][4], wlo[4][2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0[8], v1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2 v = *reinterpret_cast<const float2*>(
                    WhhT + (long)(32 * kk + 8 * s + j) * RG4 + g * RH + ub);
                v0[j] = v.x;
                v1[j] = v.y;
            }
            split8(v0, whi[g][0][kk], wlo[g][0][kk]);
            split8(v1, whi[g][1][kk], wlo[g][1][kk]);
        }
    }
    // h of a row's two adjacent units -> the tile the next step reads its A operands from
    auto put_h = [&](int buf, int lr, float2 v) {
        unsigned hi, lo;
        split_pair(v.x, v.y, hi, lo);
        *reinterpret_cast<unsigned*>(&hbuf[buf].hi[lr * RLDB + ub]) = hi;
        *reinterpret_cast<unsigned*>(&hbuf[buf].lo[lr * RLDB + ub]) = lo;
    };
#else
    // this wavefront's slice of the recurrent matrix: w[gate][hf][kb] multiplies
    // h[., 4 kb + s] into the gate pre-activation of unit ub + hf
    float w[4][2][32];
#pragma unroll
    for (int kb = 0; kb < 32; ++kb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float2 v =
                *reinterpret_cast<const float2*>(WhhT + (long)(4 * kb + s) * RG4 + g * RH + ub);
            w[g][0][kb] = v.x;
            w[g][1][kb] = v.y;
        }
    }
    auto put_h = [&](int buf, int lr, float2 v) {
        *reinterpret_cast<float2*>(&hbuf[buf][lr * RLDA + ub]) = v;
    };
#endif
#ifdef KVAE_LSTM_TILE_LOOP
    // The workgroup works through tiles blockIdx.x, blockIdx.x + gridDim.x, ... with the
    // recurrent matrix loaded once.  A tile's carried state and its first step's inputs are
    // requested while the previous tile computes (hv / cv / gx below).
    const int tile_stride = 16 * gridDim.x;
    float2 hv[4], cv[4];
    auto load_state = [&](int rb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rb + 4 * s + i;
            hv[i] = cv[i] = make_float2(0.f, 0.f);
            if (c_init && (FULL || row < B))
                cv[i] = *reinterpret_cast<const float2*>(c_init + (long)row * RH + ub);
            if (h_init && (FULL || row < B))
                hv[i] = *reinterpret_cast<const float2*>(h_init + (long)row * RH + ub);
        }
    };
    load_state(16 * blockIdx.x);
#else
    float cst[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lr = 4 * s + i, row = r0 + lr;
        cst[i][0] = cst[i][1] = 0.f;
        if (c_init && (FULL || row < B)) {
            const float2 v = *reinterpret_cast<const float2*>(c_init + (long)row * RH + ub);
            cst[i][0] = v.x;
            cst[i][1] = v.y;
        }
        // h of "step -1" (zero unless a state is carried in) goes where step 0 looks for it
        float2 v = {0.f, 0.f};
        if (h_init && (FULL || row < B))
            v = *reinterpret_cast<const float2*>(h_init + (long)row * RH + ub);
        put_h(1, lr, v);
    }
    __syncthreads();
#endif

    float wx[4][2][P > 0 ? P : 1], bs[4][2];
    if (P > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                bs[g][hf] = bias[g * RH + ub + hf];
#pragma unroll
                for (int q = 0; q < P; ++q) wx[g][hf][q] = Wih[(long)(g * RH + ub + hf) * P + q];
            }
        }
    }
    // input projection of a step for this lane's rows and units
#ifdef KVAE_LSTM_TILE_LOOP
    auto load_gx_at = [&](int r0, int t, GateRows& gx) {
#else
    auto load_gx = [&](int t, GateRows& gx) {
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * s + i;
            if (P > 0) {
                float xv[P > 0 ? P : 1];
#pragma unroll
                for (int q = 0; q < P; ++q) xv[q] = (FULL || row < B) ? x[((long)t * B + row) * P + q] : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float a = bs[g][0], b = bs[g][1];
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        a += xv[q] * wx[g][0][q];
                        b += xv[q] * wx[g][1][q];
                    }
                    gx.v[i][g] = make_float2(a, b);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    gx.v[i][g] = make_float2(0.f, 0.f);
                    if (FULL || row < B)
                        gx.v[i][g] = load_streaming(gates + t * BG + (long)row * RG4 + g * RH + ub);
                }
            }
        }
    };

    GateRows gx;
#ifdef KVAE_LSTM_TILE_LOOP
    load_gx_at(16 * blockIdx.x, 0, gx);
    for (int r0 = 16 * blockIdx.x; r0 < B; r0 += tile_stride) {
    const bool has_next = r0 + tile_stride < B;
    float cst[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cst[i][0] = cv[i].x;
        cst[i][1] = cv[i].y;
        put_h(1, 4 * s + i, hv[i]);
    }
    // LDS only: a full barrier would also wait for the previous tile's stores to land (and
    // for the prefetched loads), once per tile
    lds_barrier();
    if (has_next) load_state(r0 + tile_stride);
    auto load_gx = [&](int t, GateRows& g_) { load_gx_at(r0, t, g_); };
#else
    load_gx(0, gx);
#endif
    for (int t = 0; t < T; ++t) {
        f32x4 acc[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[g][0][i] = gx.v[i][g].x;
                acc[g][1][i] = gx.v[i][g].y;
            }
        }
        // the next step's input projection does not depend on the recurrence: its loads
        // are in flight while this step's matrix instructions run
        // (the last step loads its own inputs again rather than branch)
#ifdef KVAE_LSTM_TILE_LOOP
        if (t + 1 < T) {
            load_gx(t + 1, gx);
        } else if (has_next) {
            load_gx_at(r0 + tile_stride, 0, gx);      // the next tile's first step
        }
#else
        load_gx(t + 1 < T ? t + 1 : t, gx);
#endif
        float2 res[4][6];
        auto activate = [&](int i, int hf) {
#ifdef KVAE_EXP_NOACT
            const float a_i = acc[0][hf][i], a_f = acc[1][hf][i], a_g = acc[2][hf][i], a_o = acc[3][hf][i];
            const float cn = a_f * cst[i][hf] + a_i * a_g;
            cst[i][hf] = cn;
            const float hn = a_o * cn * 1e-3f;
#else
            const float a_i = sigmoidf_(acc[0][hf][i]), a_f = sigmoidf_(acc[1][hf][i]);
            const float a_g = tanhf_(acc[2][hf][i]), a_o = sigmoidf_(acc[3][hf][i]);
            const float cn = a_f * cst[i][hf] + a_i * a_g;
            cst[i][hf] = cn;
            const float hn = a_o * tanhf_(cn);
#endif
            if (hf == 0) {
                res[i][0].x = a_i; res[i][1].x = a_f; res[i][2].x = a_g; res[i][3].x = a_o;
                res[i][4].x = hn; res[i][5].x = cn;
            } else {
                res[i][0].y = a_i; res[i][1].y = a_f; res[i][2].y = a_g; res[i][3].y = a_o;
                res[i][4].y = hn; res[i][5].y = cn;
            }
        };
#ifdef KVAE_LSTM_BF16X3
        {
            // the step's A operands: row cc of the h tile, elements 32 kk + 8 s .. + 7, hi and lo
            const int a_off = cc * RLDB + 8 * s;
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                ah[kk] = *reinterpret_cast<const bf16x8*>(&hbuf[(t + 1) & 1].hi[a_off + 32 * kk]);
                al[kk] = *reinterpret_cast<const bf16x8*>(&hbuf[(t + 1) & 1].lo[a_off + 32 * kk]);
            }
            __builtin_amdgcn_sched_barrier(0);
#define KVAE_X3(HF_, KK_)                                                                          \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[g][HF_] =                                    \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[KK_], whi[g][HF_][KK_], acc[g][HF_], 0, 0, 0);  \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[g][HF_] =                                    \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[KK_], whi[g][HF_][KK_], acc[g][HF_], 0, 0, 0);  \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[g][HF_] =                                    \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[KK_], wlo[g][HF_][KK_], acc[g][HF_], 0, 0, 0)
            // first unit of each pair
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { KVAE_X3(0, kk); }
            // second unit, with the gate activations of the first unit between the groups
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                KVAE_X3(1, kk);
                activate(kk, 0);
            }
#undef KVAE_X3
        }
#else
        {
            // all of the step's A operands are requested before the first matrix instruction
            // (the compiler would otherwise sink each LDS read to its use and expose the
            // read latency once per group of matrix instructions)
            const float* a_row = hbuf[(t + 1) & 1] + cc * RLDA + s;
            float a[32];
#pragma unroll
            for (int kb = 0; kb < 32; ++kb) a[kb] = a_row[4 * kb];
            __builtin_amdgcn_sched_barrier(0);
            // first unit of each pair
#pragma unroll
            for (int kb = 0; kb < 32; ++kb) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb], w[g][0][kb], acc[g][0], 0, 0, 0);
            }
            // second unit, with the gate activations of the first unit between the groups
#pragma unroll
            for (int kb = 0; kb < 32; ++kb) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb], w[g][1][kb], acc[g][1], 0, 0, 0);
                if ((kb & 7) == 3) activate(kb >> 3, 0);
            }
        }
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) activate(i, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lr = 4 * s + i, row = r0 + lr;
            put_h(t & 1, lr, res[i][4]);
#ifndef KVAE_EXP_NOSTORE
            if (FULL || row < B) {
                const long o = (long)row * RH + ub;
                store_streaming(h + t * BH + o, res[i][4]);
                store_streaming(c + t * BH + o, res[i][5]);
                if (GATES) {
                    float* gr = gates + t * BG + (long)row * RG4 + ub;
                    store_streaming(gr, res[i][0]);
                    store_streaming(gr + RH, res[i][1]);
                    store_streaming(gr + 2 * RH, res[i][2]);
                    store_streaming(gr + 3 * RH, res[i][3]);
                }
            }
#endif
        }
        // orders this step's h writes before the next step's reads; the buffer written in
        // step t+1 was last read in step t, before this barrier
#ifndef KVAE_EXP_NOBARRIER
        lds_barrier();
#endif
    }
#ifdef KVAE_LSTM_TILE_LOOP
    }
#endif
}

// kernel and launcher names of the four builds of this file
#ifdef KVAE_LSTM_BF16X3
#ifdef KVAE_LSTM_TILE_LOOP
#define KVAE_FWD_KERNEL lstm128_split_forward_tiles_kernel
#else
#define KVAE_FWD_KERNEL lstm128_split_forward_kernel
#endif
#define KVAE_BWD_KERNEL lstm128_split_backward_kernel
#define KVAE_LAUNCH_TILES launch_lstm128_forward_tiles_split
#define KVAE_LAUNCH_FWD launch_lstm128_forward_split
#define KVAE_LAUNCH_BWD launch_lstm128_backward_split
#else
#ifdef KVAE_LSTM_TILE_LOOP
#define KVAE_FWD_KERNEL lstm128_resident_forward_tiles_kernel
#else
#define KVAE_FWD_KERNEL lstm128_resident_forward_kernel
#endif
#define KVAE_BWD_KERNEL lstm128_resident_backward_kernel
#define KVAE_LAUNCH_TILES launch_lstm128_forward_tiles
#define KVAE_LAUNCH_FWD launch_lstm128_forward_exact
#define KVAE_LAUNCH_BWD launch_lstm128_backward_exact
#endif
template <int P>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
KVAE_FWD_KERNEL(int T, int B, const float* __restrict__ x,
                                const float* __restrict__ Wih, const float* __restrict__ bias,
                                const float* __restrict__ WhhT,   // (H,4H)
                                float* gates, float* __restrict__ h, float* __restrict__ c,
                                const float* __restrict__ h_init,  // (B,H) or null
                                const float* __restrict__ c_init, int store_gates) {
#ifdef KVAE_LSTM_BF16X3
    __shared__ __attribute__((aligned(16))) HTile hbuf[2];
#else
    __shared__ __attribute__((aligned(16))) float hbuf[2][16 * RLDA];
#endif
#ifdef KVAE_LSTM_TILE_LOOP
    const bool full = (B & 15) == 0;
#else
    const bool full = (int)blockIdx.x * 16 + 16 <= B;
#endif
#define KVAE_FWD_BODY(F_, G_) \
    lstm128_forward_body<P, F_, G_>(T, B, x, Wih, bias, WhhT, gates, h, c, h_init, c_init, hbuf)
    if (store_gates) {
        if (full) KVAE_FWD_BODY(true, true); else KVAE_FWD_BODY(false, true);
    } else {
        if (full) KVAE_FWD_BODY(true, false); else KVAE_FWD_BODY(false, false);
    }
#undef KVAE_FWD_BODY
}

#ifdef KVAE_LSTM_TILE_LOOP
// This translation unit is compiled a second time with KVAE_LSTM_TILE_LOOP for the
// many-tiles-per-workgroup form of the forward kernel alone (in one translation unit the
// two forms disturb each other's register allocation, and the single-tile form has none to
// spare).
int KVAE_LAUNCH_TILES(int T, int B, int P, int workgroups, const float* x,
                      const float* Wih, const float* bias, const float* WhhT,
                      float* gates, float* h, float* c, const float* h_init,
                      const float* c_init, int store_gates, hipStream_t stream) {
    const dim3 grid(workgroups);
#define KVAE_R128T(P_)                                                                       \
    hipLaunchKernelGGL(KVAE_FWD_KERNEL<P_>, grid, dim3(256), 0, stream,                      \
                       T, B, x, Wih, bias, WhhT, gates, h, c, h_init, c_init, store_gates)
    switch (P) {
        case 0: KVAE_R128T(0); break;
        case 1: KVAE_R128T(1); break;
        case 2: KVAE_R128T(2); break;
        case 3: KVAE_R128T(3); break;
        case 4: KVAE_R128T(4); break;
        default: return KVAE_E_UNSUPPORTED;
    }
#undef KVAE_R128T
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // namespace kvae
#else   // the rest of the file: single-tile translation unit only

// what the cell adjoint of one step reads from global memory for this lane's 4 rows x 2
// units; independent of the recurrence, so loaded a step ahead
template <int KH>
struct CellRows {
    float2 ig[4], fg[4], gg[4], og[4], cc[4], cp[4], gh[4];
    // softmax output and its gradient: lane (lane & 3) = k of every quad holds entry k of
    // each of the lane's four rows (every lane of a row group would hold the same KH values)
    float al[4], ga[4];
};

// The softmax head on top of the layer, alpha = softmax(h W^T + b) with W (KH,H)
// (kvae/dynamics_parameter_network.py:57-58): with KH > 0 the kernel forms the gradient
// w.r.t. h_t from alpha_t and its gradient instead of reading it, and sums dW, db.
// alpha, g_alpha: (T,B,KH); partial[blockIdx.x][ KH*H | KH ].
struct HeadAdjoint {
    const float* alpha;
    const float* g_alpha;
    const float* W;
    float* partial;
};

// grid = ceil(B/16), block = 256.  g_h (T,B,H): gradient w.r.t. every h_t from above.
// Whh (4H,H) as stored.  Outputs: dG (T,B,4H) gradient w.r.t. the gate pre-activations;
// partial[blockIdx.x][ 4H | 4H | 4H*P ]: per-workgroup sums over (t, rows) of dG (twice:
// bias_ih and bias_hh each get memory of their own) and, for P > 0, of dG^T x (the input
// weight gradient); for P > 0 also dx (T,B,P) = dG Wih.
template <int P, bool FULL, int KH>
__device__ __forceinline__ void
lstm128_backward_body(int T, int B, const float* __restrict__ x, const float* __restrict__ Wih,
                      const float* __restrict__ gates, const float* __restrict__ cg,
                      const float* __restrict__ c_init, const float* __restrict__ g_h,
                      const float* __restrict__ Whh, float* __restrict__ dG,
                      float* __restrict__ dx, float* __restrict__ partial, const HeadAdjoint head,
                      KVAE_GTILE_T, float (&dx_part)[2][4][16][P > 0 ? P : 1]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cc = lane & 15, s = lane >> 4;
    const int r0 = blockIdx.x * 16, u0 = 32 * wave, ub = u0 + 2 * cc;
    const long BH = (long)B * RH, BG = (long)B * RG4;

#ifdef KVAE_LSTM_BF16X3
    // w*[hf][kk][j] multiplies dG[., 32 kk + 8 s + j] into dh of unit ub + hf
    bf16x8 whi[2][16], wlo[2][16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        float v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 v =
                *reinterpret_cast<const float2*>(Whh + (long)(32 * kk + 8 * s + j) * RH + ub);
            v0[j] = v.x;
            v1[j] = v.y;
        }
        split8(v0, whi[0][kk], wlo[0][kk]);
        split8(v1, whi[1][kk], wlo[1][kk]);
    }
#else
    // w[hf][kb] multiplies dG[., 4 kb + s] into dh of unit ub + hf
    float w[2][128];
#pragma unroll
    for (int kb = 0; kb < 128; ++kb) {
        const float2 v = *reinterpret_cast<const float2*>(Whh + (long)(4 * kb + s) * RH + ub);
        w[0][kb] = v.x;
        w[1][kb] = v.y;
    }
#endif
    float dc[4][2], sb[4][2], sw[4][2][P > 0 ? P : 1], wx[4][2][P > 0 ? P : 1];
    f32x4 rec[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dc[i][0] = dc[i][1] = 0.f;
        rec[0][i] = rec[1][i] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            sb[g][hf] = 0.f;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                sw[g][hf][q] = 0.f;
                wx[g][hf][q] = Wih[(long)(g * RH + ub + hf) * P + q];
            }
        }
    }
    // c_t of step t is c_{t-1} of step t+1, which the previous iteration already holds: only
    // the first step (t = T-1) reads it from memory
    float wh[KH > 0 ? KH : 1][2], hw[KH > 0 ? KH : 1][2], hb = 0.f;
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        wh[k][0] = head.W[k * RH + ub];
        wh[k][1] = head.W[k * RH + ub + 1];
        hw[k][0] = hw[k][1] = 0.f;
    }
    // first: the step the sweep starts with (t = T-1), the only one that reads its own c_t
    // (std::true_type / std::false_type).  FULL: every access is unconditional and the address
    // of c_{t-1} is chosen by scalar selects (the step below's c; c_init at t = 0; any valid
    // address times zero without one), so the time loop below has no branch and the compiler
    // counts the outstanding memory operations exactly -- with branches it waits for ALL of
    // them, the step's 16 stores of gate gradients included, before it touches the next step's
    // inputs.
    auto load_rows = [&](int t, auto first, CellRows<KH>& in) {
        const float2 z = {0.f, 0.f};
        constexpr bool FIRST = decltype(first)::value;
        const float* cp_base = t > 0 ? cg + (long)(t - 1) * BH : (c_init ? c_init : cg);
        const float cp_scale = (t > 0 || c_init) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * s + i;
            const float2 c_here = in.cp[i];
            const long o = (long)row * RH + ub;
            if constexpr (FULL) {
                const float* gr = gates + t * BG + (long)row * RG4 + ub;
                in.ig[i] = load_streaming(gr);
                in.fg[i] = load_streaming(gr + RH);
                in.gg[i] = load_streaming(gr + 2 * RH);
                in.og[i] = load_streaming(gr + 3 * RH);
                in.cc[i] = FIRST ? load_streaming(cg + t * BH + o) : c_here;
                const float2 cp = load_streaming(cp_base + o);
                in.cp[i] = make_float2(cp.x * cp_scale, cp.y * cp_scale);
                in.gh[i] = KH == 0 ? load_streaming(g_h + t * BH + o) : z;
                if (KH > 0) {
                    // (lanes whose k is not a logit read logit 0 and drop it)
                    const int k = cc & 3, kq = k < KH ? k : 0;
                    const float al = head.alpha[((long)t * B + row) * KH + kq];
                    const float ga = head.g_alpha[((long)t * B + row) * KH + kq];
                    in.al[i] = k < KH ? al : 0.f;
                    in.ga[i] = k < KH ? ga : 0.f;
                }
            } else {
                in.ig[i] = in.fg[i] = in.gg[i] = in.og[i] = in.cp[i] = in.gh[i] = z;
                in.cc[i] = FIRST ? z : c_here;
                if (row < B) {
                    const float* gr = gates + t * BG + (long)row * RG4 + ub;
                    in.ig[i] = load_streaming(gr);
                    in.fg[i] = load_streaming(gr + RH);
                    in.gg[i] = load_streaming(gr + 2 * RH);
                    in.og[i] = load_streaming(gr + 3 * RH);
                    if (FIRST) in.cc[i] = load_streaming(cg + t * BH + o);
                    if (t > 0) {
                        in.cp[i] = load_streaming(cg + (t - 1) * BH + o);
                    } else if (c_init) {
                        in.cp[i] = *reinterpret_cast<const float2*>(c_init + o);
                    }
                    if (KH == 0) in.gh[i] = load_streaming(g_h + t * BH + o);
                }
                if (KH > 0) {
                    const int k = cc & 3;
                    const bool ok = row < B && k < KH;
                    in.al[i] = ok ? head.alpha[((long)t * B + row) * KH + k] : 0.f;
                    in.ga[i] = ok ? head.g_alpha[((long)t * B + row) * KH + k] : 0.f;
                }
            }
        }
    };

    CellRows<KH> in;
#pragma unroll
    for (int i = 0; i < 4; ++i) in.cp[i] = make_float2(0.f, 0.f);
    load_rows(T - 1, std::true_type{}, in);
    // one step; more: whether steps below it follow (a bool, or std::true_type /
    // std::false_type where the caller knows: then nothing in the step branches on it)
    auto step = [&](int t, auto more) {
#ifndef KVAE_LSTM_BF16X3
        float* g_tile = tile[t & 1];
#endif
        float2 dgs[4][4];   // this step's gate gradients, stored during the matrix loop
        float xv[4][P > 0 ? P : 1];
        if (P > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = r0 + 4 * s + i;
#pragma unroll
                for (int q = 0; q < P; ++q) xv[i][q] = (FULL || row < B) ? x[((long)t * B + row) * P + q] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lr = 4 * s + i;
            float2 gh = in.gh[i];
            const float tc0 = tanhf_(in.cc[i].x), tc1 = tanhf_(in.cc[i].y);
            if (KH > 0) {
                // softmax adjoint of this row, then d h = dl W; dW += dl^T h with h = o tanh(c)
                float dot = in.al[i] * in.ga[i];           // sum over the quad's lanes = over k
                dot += row_exchange16<0xB1>(dot);
                dot += row_exchange16<0x4E>(dot);
                const float dl_own = in.al[i] * (in.ga[i] - dot);
                hb += dl_own;
                const float h0v = in.og[i].x * tc0, h1v = in.og[i].y * tc1;
                gh = make_float2(0.f, 0.f);
                const float dl[4] = {row_exchange16<0x00>(dl_own), row_exchange16<0x55>(dl_own),
                                     row_exchange16<0xAA>(dl_own), row_exchange16<0xFF>(dl_own)};
#pragma unroll
                for (int k = 0; k < KH; ++k) {
                    gh.x += dl[k] * wh[k][0];
                    gh.y += dl[k] * wh[k][1];
                    hw[k][0] += dl[k] * h0v;
                    hw[k][1] += dl[k] * h1v;
                }
            }
            const float dh0 = rec[0][i] + gh.x, dh1 = rec[1][i] + gh.y;
            float2 di, df, dg, dout;
            {
                const float tc = tc0;
                const float dct = dh0 * in.og[i].x * (1.f - tc * tc) + dc[i][0];
                di.x = dct * in.gg[i].x * in.ig[i].x * (1.f - in.ig[i].x);
                df.x = dct * in.cp[i].x * in.fg[i].x * (1.f - in.fg[i].x);
                dg.x = dct * in.ig[i].x * (1.f - in.gg[i].x * in.gg[i].x);
                dout.x = dh0 * tc * in.og[i].x * (1.f - in.og[i].x);
                dc[i][0] = dct * in.fg[i].x;
            }
            {
                const float tc = tc1;
                const float dct = dh1 * in.og[i].y * (1.f - tc * tc) + dc[i][1];
                di.y = dct * in.gg[i].y * in.ig[i].y * (1.f - in.ig[i].y);
                df.y = dct * in.cp[i].y * in.fg[i].y * (1.f - in.fg[i].y);
                dg.y = dct * in.ig[i].y * (1.f - in.gg[i].y * in.gg[i].y);
                dout.y = dh1 * tc * in.og[i].y * (1.f - in.og[i].y);
                dc[i][1] = dct * in.fg[i].y;
            }
            if (KH > 0) {
                // with the head's accumulators in the kernel the registers for deferred
                // stores are not there: the row's gate gradients go out at once
                if (FULL || r0 + lr < B) {
                    float* go = dG + t * BG + (long)(r0 + lr) * RG4 + ub;
                    store_streaming(go, di);
                    store_streaming(go + RH, df);
                    store_streaming(go + 2 * RH, dg);
                    store_streaming(go + 3 * RH, dout);
                }
            } else {
                dgs[i][0] = di;
                dgs[i][1] = df;
                dgs[i][2] = dg;
                dgs[i][3] = dout;
            }
            const float2 dgt[4] = {di, df, dg, dout};
#ifdef KVAE_LSTM_BF16X3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned hi, lo;
                split_pair(dgt[g].x, dgt[g].y, hi, lo);
                *reinterpret_cast<unsigned*>(&tile[t & 1].hi[lr * RLDGB + g * RH + ub]) = hi;
                *reinterpret_cast<unsigned*>(&tile[t & 1].lo[lr * RLDGB + g * RH + ub]) = lo;
            }
#else
            float* tl = g_tile + lr * RLDG + ub;
            *reinterpret_cast<float2*>(tl) = di;
            *reinterpret_cast<float2*>(tl + RH) = df;
            *reinterpret_cast<float2*>(tl + 2 * RH) = dg;
            *reinterpret_cast<float2*>(tl + 3 * RH) = dout;
#endif
            float dxp[P > 0 ? P : 1];
#pragma unroll
            for (int q = 0; q < P; ++q) dxp[q] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sb[g][0] += dgt[g].x;
                sb[g][1] += dgt[g].y;
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    sw[g][0][q] += dgt[g].x * xv[i][q];
                    sw[g][1][q] += dgt[g].y * xv[i][q];
                    dxp[q] += dgt[g].x * wx[g][0][q] + dgt[g].y * wx[g][1][q];
                }
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const float v = row_sum16(dxp[q]);
                if (cc == 0) dx_part[t & 1][wave][lr][q] = v;
            }
        }
        // inputs of the next step, in flight during the matrix instructions below
        if (more) load_rows(t - 1, std::false_type{}, in);
        lds_barrier();
        if (P > 0 && threadIdx.x < 16 * P) {
            const int lr = threadIdx.x / (P > 0 ? P : 1), q = threadIdx.x % (P > 0 ? P : 1);
            if (FULL || r0 + lr < B)
                dx[((long)t * B + r0 + lr) * P + q] =
                    dx_part[t & 1][0][lr][q] + dx_part[t & 1][1][lr][q] +
                    dx_part[t & 1][2][lr][q] + dx_part[t & 1][3][lr][q];
        }
        // dh_{t-1} (recurrent part) = dG_t Whh
#pragma unroll
        for (int i = 0; i < 4; ++i) rec[0][i] = rec[1][i] = 0.f;
        // the step's gate gradients go to global memory one store per group of matrix
        // instructions (issued together, every compute unit would write at the same moment
        // and wait for memory while the matrix pipe idles)
        auto store_dg = [&](int idx) {
            if (KH > 0) return;
            const int i = idx >> 2, g = idx & 3;
            if (FULL || r0 + 4 * s + i < B) store_streaming(dG + t * BG + (long)(r0 + 4 * s + i) * RG4 + g * RH + ub, dgs[i][g]);
        };
#ifdef KVAE_LSTM_BF16X3
        if (more) {
            // A operands: row cc of the gate-gradient tile, elements 32 kk + 8 s .. + 7; read
            // AG k-groups ahead of the matrix instructions that use them (one with the softmax
            // head's accumulators in the kernel, which needs the registers)
            constexpr int AG = KH > 0 ? 1 : 2;
            const int a_off = cc * RLDGB + 8 * s;
            bf16x8 ah[2][AG], al[2][AG];
            auto read_group = [&](int grp, bf16x8 (&h_)[AG], bf16x8 (&l_)[AG]) {
#pragma unroll
                for (int q = 0; q < AG; ++q) {
                    h_[q] = *reinterpret_cast<const bf16x8*>(&tile[t & 1].hi[a_off + 32 * (AG * grp + q)]);
                    l_[q] = *reinterpret_cast<const bf16x8*>(&tile[t & 1].lo[a_off + 32 * (AG * grp + q)]);
                }
            };
            read_group(0, ah[0], al[0]);
#pragma unroll
            for (int grp = 0; grp < 16 / AG; ++grp) {
                if (grp + 1 < 16 / AG) read_group(grp + 1, ah[(grp + 1) & 1], al[(grp + 1) & 1]);
#pragma unroll
                for (int q = 0; q < AG; ++q) {
                    const int kk = AG * grp + q;
                    rec[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], whi[0][kk], rec[0], 0, 0, 0);
                    rec[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], whi[1][kk], rec[1], 0, 0, 0);
                    rec[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[grp & 1][q], whi[0][kk], rec[0], 0, 0, 0);
                    rec[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[grp & 1][q], whi[1][kk], rec[1], 0, 0, 0);
                    rec[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], wlo[0][kk], rec[0], 0, 0, 0);
                    rec[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], wlo[1][kk], rec[1], 0, 0, 0);
                    store_dg(kk);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#else
        if (more) {
            const float* a_row = g_tile + cc * RLDG + s;
#pragma unroll
            for (int kb = 0; kb < 128; ++kb) {
                const float a = a_row[4 * kb];
                rec[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[0][kb], rec[0], 0, 0, 0);
                rec[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[1][kb], rec[1], 0, 0, 0);
                if ((kb & 7) == 7) {
                    store_dg(kb >> 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#endif
#pragma unroll
            for (int idx = 0; idx < 16; ++idx) store_dg(idx);
        }
        // tile[t & 1] is rewritten in step t-2, after the barrier of step t-1 that every
        // wavefront reaches only after its reads above
    };
    if constexpr (FULL) {
        for (int t = T - 1; t >= 1; --t) step(t, std::true_type{});
        step(0, std::false_type{});
    } else {
        for (int t = T - 1; t >= 0; --t) step(t, t > 0);
    }
    if (KH > 0) {
        // every lane of the workgroup that owns row group s saw the same dl: the weight sums
        // are per unit (lanes with s == 0 write after adding the four row groups), the bias
        // sums are taken from one lane per row group
        float* hp = head.partial + (long)blockIdx.x * (KH * (RH + 1));
#pragma unroll
        for (int k = 0; k < KH; ++k) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float v = hw[k][hf];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (s == 0) hp[k * RH + ub + hf] = v;
            }
        }
        // lane k of a quad summed d logit_k of its rows
        float b = hb;
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (threadIdx.x < KH) hp[KH * RH + threadIdx.x] = b;
    }
    // sum the per-lane column accumulators over the four row groups (lane bits 4, 5)
    float* part = partial + (long)blockIdx.x * (RG4 * (2 + P));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int col = g * RH + ub + hf;
            float a = sb[g][hf];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            if (s == 0) {
                part[col] = a;
                part[RG4 + col] = a;
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
                float v = sw[g][hf][q];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (s == 0) part[2 * RG4 + (long)col * P + q] = v;
            }
        }
    }
}

template <int P, int KH, bool FULL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
KVAE_BWD_KERNEL(int T, int B, const float* __restrict__ x,
                                 const float* __restrict__ Wih, const float* __restrict__ gates,
                                 const float* __restrict__ cg, const float* __restrict__ c_init,
                                 const float* __restrict__ g_h, const float* __restrict__ Whh,
                                 float* __restrict__ dG, float* __restrict__ dx,
                                 float* __restrict__ partial, const HeadAdjoint head) {
#ifdef KVAE_LSTM_BF16X3
    __shared__ __attribute__((aligned(16))) GTile tile[2];
#else
    __shared__ __attribute__((aligned(16))) float tile[2][16 * RLDG];
#endif
    __shared__ float dx_part[2][4][16][P > 0 ? P : 1];
    // (full and ragged tiles are separate kernels: an unpredicated path beside the predicated
    // one in the same kernel costs registers the resident matrix needs)
    lstm128_backward_body<P, FULL, KH>(T, B, x, Wih, gates, cg, c_init, g_h, Whh, dG, dx,
                                        partial, head, tile, dx_part);
}

// single-tile launchers of this build (the C entry points below pick a build)
int KVAE_LAUNCH_FWD(int T, int B, int P, const float* x, const float* Wih, const float* bias,
                    const float* WhhT, float* gates, float* h, float* c, const float* h_init,
                    const float* c_init, int store_gates, hipStream_t stream) {
    const dim3 grid((B + 15) / 16);
#define KVAE_R128F(P_)                                                                   \
    hipLaunchKernelGGL(KVAE_FWD_KERNEL<P_>, grid, dim3(256), 0, stream, T, B, x, Wih, bias, \
                       WhhT, gates, h, c, h_init, c_init, store_gates)
    switch (P) {
        case 0: KVAE_R128F(0); break;
        case 1: KVAE_R128F(1); break;
        case 2: KVAE_R128F(2); break;
        case 3: KVAE_R128F(3); break;
        case 4: KVAE_R128F(4); break;
        default: return KVAE_E_UNSUPPORTED;
    }
#undef KVAE_R128F
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int KVAE_LAUNCH_BWD(int T, int B, int P, const float* x, const float* Wih, const float* gates,
                    const float* c, const float* c_init, const float* g_h, const float* Whh,
                    float* dG, float* dx, float* partial, int KH, const float* alpha,
                    const float* g_alpha, const float* Wh, float* head_partial,
                    hipStream_t stream) {
    const dim3 grid((B + 15) / 16);
    const HeadAdjoint head = {alpha, g_alpha, Wh, head_partial};
    const bool full = B % 16 == 0;
    if (KH > 0) {
        // the head sits on the upper layer, whose input projection is never formed in the kernel
        if (P != 0) return KVAE_E_UNSUPPORTED;
        // (with the head's accumulators in the kernel the branch-free instance measured slower,
        // 138 against 103 us at T = 20, B = 4096, with immediate and with deferred stores of the
        // gate gradients alike: these kernels keep the predicated form)
#define KVAE_R128H(K_)                                                                        \
    case K_:                                                                                  \
        hipLaunchKernelGGL((KVAE_BWD_KERNEL<0, K_, false>), grid, dim3(256), 0, stream, T, B, \
                           x, Wih, gates, c, c_init, g_h, Whh, dG, dx, partial, head);        \
        break;
        switch (KH) {
            KVAE_R128H(2) KVAE_R128H(3) KVAE_R128H(4)
            default: return KVAE_E_UNSUPPORTED;
        }
#undef KVAE_R128H
        return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
    }
#define KVAE_R128B(P_)                                                                         \
    do {                                                                                       \
        if (full)                                                                              \
            hipLaunchKernelGGL((KVAE_BWD_KERNEL<P_, 0, true>), grid, dim3(256), 0, stream, T, B, \
                               x, Wih, gates, c, c_init, g_h, Whh, dG, dx, partial, head);     \
        else                                                                                   \
            hipLaunchKernelGGL((KVAE_BWD_KERNEL<P_, 0, false>), grid, dim3(256), 0, stream, T, B, \
                               x, Wih, gates, c, c_init, g_h, Whh, dG, dx, partial, head);     \
    } while (0)
    switch (P) {
        case 0: KVAE_R128B(0); break;
        case 1: KVAE_R128B(1); break;
        case 2: KVAE_R128B(2); break;
        case 3: KVAE_R128B(3); break;
        case 4: KVAE_R128B(4); break;
        default: return KVAE_E_UNSUPPORTED;
    }
#undef KVAE_R128B
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // namespace kvae

#ifndef KVAE_LSTM_BF16X3
using namespace kvae;

// workgroups that are resident at once: one per compute unit (each takes all registers)
static int resident_workgroups() { return device_compute_units(); }

// 0: exact fp32 matrix instructions; 1: three bf16 instructions per product (default).
// KVAE_LSTM_PRECISION=fp32 in the environment selects 0 at the first call.
static int g_split = -1;
static int split_products() {
    if (g_split < 0) {
        const char* e = getenv("KVAE_LSTM_PRECISION");
        g_split = (e && (!strcmp(e, "fp32") || !strcmp(e, "exact"))) ? 0 : 1;
    }
    return g_split;
}

extern "C" {

int kvae_lstm128_blocks(int B) { return B < 1 ? 0 : (B + 15) / 16; }

int kvae_lstm128_set_precision(int split) {
    if (split != 0 && split != 1) return KVAE_E_BADARG;
    g_split = split;
    return KVAE_OK;
}

int kvae_lstm128_get_precision(void) { return split_products(); }

int kvae_lstm128_forward(int T, int B, int P, const void* x, const void* Wih, const void* bias,
                         const void* WhhT, void* gates, void* h, void* c, const void* h_init,
                         const void* c_init, int store_gates, void* stream) {
    if (T < 1 || B < 1 || !WhhT || !h || !c || (h_init == nullptr) != (c_init == nullptr) ||
        (P > 0 && (!x || !Wih || !bias)) || (P == 0 && !gates) || (store_gates && !gates))
        return KVAE_E_BADARG;
    const bool split = split_products() != 0;
    // A workgroup loads its 256 KB of the recurrent matrix once.  Over a whole sequence that
    // is nothing; a one-step call over very many rows (the masked filter steps the network
    // once per time step) would spend its time on those loads, so it gets one workgroup
    // per compute unit, each working through many tiles.
    if (T < 8 && (B + 15) / 16 > 2 * resident_workgroups())
        return (split ? launch_lstm128_forward_tiles_split : launch_lstm128_forward_tiles)(
            T, B, P, resident_workgroups(), (const float*)x, (const float*)Wih,
            (const float*)bias, (const float*)WhhT, (float*)gates, (float*)h, (float*)c,
            (const float*)h_init, (const float*)c_init, store_gates, (hipStream_t)stream);
    return (split ? launch_lstm128_forward_split : launch_lstm128_forward_exact)(
        T, B, P, (const float*)x, (const float*)Wih, (const float*)bias, (const float*)WhhT,
        (float*)gates, (float*)h, (float*)c, (const float*)h_init, (const float*)c_init,
        store_gates, (hipStream_t)stream);
}

int kvae_lstm128_backward(int T, int B, int P, const void* x, const void* Wih, const void* gates,
                          const void* c, const void* c_init, const void* g_h, const void* Whh,
                          void* dG, void* dx, void* partial, void* stream) {
    if (T < 1 || B < 1 || !gates || !c || !g_h || !Whh || !dG || !partial ||
        (P > 0 && (!x || !Wih || !dx)))
        return KVAE_E_BADARG;
    return (split_products() ? launch_lstm128_backward_split : launch_lstm128_backward_exact)(
        T, B, P, (const float*)x, (const float*)Wih, (const float*)gates, (const float*)c,
        (const float*)c_init, (const float*)g_h, (const float*)Whh, (float*)dG, (float*)dx,
        (float*)partial, 0, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int kvae_lstm128_backward_head(int T, int B, int K, const void* gates, const void* c,
                               const void* c_init, const void* alpha, const void* g_alpha,
                               const void* Wh, const void* Whh, void* dG, void* partial,
                               void* head_partial, void* stream) {
    if (T < 1 || B < 1 || !gates || !c || !alpha || !g_alpha || !Wh || !Whh || !dG || !partial ||
        !head_partial)
        return KVAE_E_BADARG;
    if (K < 2 || K > 4) return KVAE_E_UNSUPPORTED;
    return (split_products() ? launch_lstm128_backward_split : launch_lstm128_backward_exact)(
        T, B, 0, nullptr, nullptr, (const float*)gates, (const float*)c, (const float*)c_init,
        nullptr, (const float*)Whh, (float*)dG, nullptr, (float*)partial, K, (const float*)alpha,
        (const float*)g_alpha, (const float*)Wh, (float*)head_partial, (hipStream_t)stream);
}

}  // extern "C"
#endif   // KVAE_LSTM_BF16X3
#endif   // KVAE_LSTM_TILE_LOOP
