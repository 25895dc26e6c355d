// ONE time step of an upper LSTM layer (hidden size 128, input = the lower layer's h, also
// 128 wide) over very many rows, input projection included:
//     gates = x W_ih^T + h_prev W_hh^T + b,   c = f c_prev + i g,   h = o tanh(c)
// (one step of torch.nn.LSTM's second layer, kvae/dynamics_parameter_network.py:46-58, as the
// masked filter calls it once per time step: kvae/state_space_model.py:279-282).
//
// Run as two launches (a row-streaming product that writes the (rows, 512) input projection,
// then the recurrent kernel that reads it back) the step moves 6.5 KB per row through HBM, 4 KB
// of it the projection's round trip.  Both matrices do not fit one compute unit's registers
// (2 x 256 KB as bf16 hi + lo halves), but a single step has no recurrence inside the
// launch, so the hidden units can be cut in two: a workgroup owns 64 units, keeps the
// 256 x 256 slice [W_ih^T ; W_hh^T] of their four gates in registers (256 per lane, B
// operands of v_mfma_f32_16x16x32_bf16, split into hi and lo bf16 halves) and works
// through 16-row tiles of [x | h_prev]: global -> registers -> split -> LDS (double
// buffered, the next tile in flight during this tile's matrix instructions) -> A
// operands, 96 matrix instructions per wavefront and tile, then the cell update in the
// epilogue.  3.5 KB per row (x and h_prev are read by both halves; the two workgroups of a
// tile sit on one XCD and walk the tiles in the same order, so the second read comes
// from L2).  Products are a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32 accumulation, as in
// lstm_resident.hip.  Wavefront w of half hh owns units 64 hh + 16 w .. + 15, lane (cc, s)
// unit 64 hh + 16 w + cc of rows 4s .. 4s+3: its four accumulators are that unit's i, f, g, o.
//
// The same kernel steps the LOWER layer (template P = its input width, 1 or 2): the narrow
// input is projected with plain multiply-adds, only h_prev W_hh^T runs on the matrix
// instruction (K = 128), so one workgroup holds all of the layer's matrix: wavefront w owns
// units 32 w .. 32 w + 31, lane (cc, s) the adjacent units 32 w + 2 cc, + 1 (8-byte stores),
// 2 KB per row.  The many-tiles form of the recurrent kernel (lstm_resident.hip) keeps one
// tile in flight and runs such a step at 2.1 TB/s; this one at the rate of the upper layer.
//
// What bounds a step (262144 rows, MI355X): the matrix instructions.  With half of them
// removed the upper layer takes 166 instead of 255 us and the lower layer 127 instead of
// 167 us; without the stores 239 / 144 us, without the exp / rcp chains of the cell update
// 227 / 150 us.  96 instructions per wavefront and tile come to about 14 ns each: the chip
// holds its clock near 1.6 GHz under this load, where the instruction's 16 cycles are 10 ns.
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "lstm_common.h"

// This file is compiled three times: as it is (the lower layer's one-input instances and the C
// entry point), with KVAE_SF_UPPER (the upper layer's instances) and with KVAE_SF_LOWER2 (the
// lower layer's two-input instances); the last two keep their weights pinned to accumulation
// registers and are built with the option that keeps matrix-instruction results in ordinary
// registers.  Those two are assembled from the compiler's listing after one wait state has been
// put between every packed vector instruction and a directly following matrix instruction: a
// packed fp32 instruction with op_sel on its second operand directly before a bf16 matrix
// instruction computes its low result wrongly in lanes 48-63 on this hardware, and the compiler
// emitted such a pair in the two-input instance (tools/exp/packed_before_mfma.py,
// tools/exp/pk_mfma_repro.hip; the Makefile has the steps).
// KVAE_SF_PIN_LOWER = 1 pins the weights of whatever lower-layer instance a build holds.  In the
// default build that was miscompiled (see the pin below); the listing of the failing kernel is
// kept as tests/golden/lstm_step_pinned_lower_gfx950.s for tools/exp/orphan_spills.py.
#ifndef KVAE_SF_PIN_LOWER
#ifdef KVAE_SF_LOWER2
#define KVAE_SF_PIN_LOWER 1
#else
#define KVAE_SF_PIN_LOWER 0
#endif
#endif

namespace kvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sf_split_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const f32x2v v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2v r = {x - __builtin_bit_cast(float, hi << 16),
                      y - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void sf_split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sf_split_pair(x[2 * j], x[2 * j + 1], h[j], l[j]);
    const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
    hi = __builtin_bit_cast(bf16x8, hv);
    lo = __builtin_bit_cast(bf16x8, lv);
}

constexpr int SF_H = 128, SF_G4 = 512;
constexpr int SF_K = 2 * SF_H;            // [x | h_prev]

// P == 0 (upper layer): grid = 16 * groups, block = 256; x (M,128), Wx = W_ih^T (128,512).
// P in 1..2 (lower layer): grid = walkers, block = 256; x (M,P), Wx = W_ih (512,P) as stored.
// h_prev, c_prev, h_out, c_out: (M,128); h_prev / c_prev may be null (zero state).
// WhhT: (128,512) = the stored (4H,H) matrix transposed; bias (512) = b_ih + b_hh.
// FULL: M is a multiple of 16 (no row predicates); STATE: h_prev / c_prev are given.
template <int P, bool FULL, bool STATE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
lstm128_step_fused_kernel(long M, const float* __restrict__ x, const float* __restrict__ Wx,
                          const float* __restrict__ h_prev, const float* __restrict__ c_prev,
                          const float* __restrict__ WhhT, const float* __restrict__ bias,
                          float* __restrict__ h_out, float* __restrict__ c_out) {
    constexpr int UB = P > 0 ? 2 : 1;            // units per lane
    constexpr int KD = P > 0 ? SF_H : SF_K;      // depth of the matrix-instruction product
    constexpr int KK = KD / 32;
    constexpr int LD = KD + 8;                   // bf16 per LDS row: rows 4 banks apart
    constexpr int V4 = KD * 16 / (256 * 4);      // float4 loads per lane per tile
    struct Tile {
        __bf16 hi[16 * LD], lo[16 * LD];
    };
    __shared__ __attribute__((aligned(16))) Tile xt[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cc = lane & 15, s = lane >> 4;
    // upper layer: consecutive workgroup ids go round the 8 XCDs, and the two halves of a tile
    // walker share one
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    const int half = P > 0 ? 0 : (j & 1);
    const long walker = P > 0 ? bid : (long)(j >> 1) * 8 + xcd;
    const long walkers = P > 0 ? gridDim.x : gridDim.x / 2;
    const int unit0 = P > 0 ? 32 * wave + 2 * cc : 64 * half + 16 * wave + cc;   // + ub

    // w*[ub][g][kk] multiplies the tile's column 32 kk + 8 s + . into gate g of unit0 + ub
    bf16x8 whi[UB][4][KK], wlo[UB][4][KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const float* W = (P == 0 && kk < 4) ? Wx : WhhT;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) {
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    v[jj] = W[(long)(32 * (kk & 3) + 8 * s + jj) * SF_G4 + g * SF_H + unit0 + ub];
                sf_split8(v, whi[ub][g][kk], wlo[ub][g][kk]);
                // the matrix instruction reads its B operand from accumulation registers as
                // well: keep the upper layer's weights there, so that the 256 ordinary
                // registers hold the tiles in flight without copies between the two files
                // (484 -> 406 instructions per tile; the lower layer's two-input instance 693 ->
                // 455).  The pin is safe only while the allocator never evicts a pinned vector:
                // with the accumulators in accumulation registers too (the default build) the
                // lower layer's 64 weight vectors fill that file, the accumulators want 32 of
                // its registers, and the allocator evicts pinned vectors.  It saves the pieces
                // of an evicted 128-bit vector in different places (two dwords to scratch, two
                // to spare registers) and restores only three of the four before the matrix
                // instruction that reads the vector: the fourth operand register keeps its
                // stale content, and the product is right or wrong with that content.
                // tools/exp/orphan_spills.py finds the unrestored piece in the listing.  (Whether
                // a pinned vector is evicted there changes with every edit of this kernel.)
                // The builds that pin keep their
                // accumulators in ordinary registers and the Makefile refuses their objects if
                // the listing has any spill; the one-input instance spills even so and stays
                // unpinned.
                if constexpr (P == 0 || KVAE_SF_PIN_LOWER)
                    asm volatile("" : "+a"(whi[ub][g][kk]), "+a"(wlo[ub][g][kk]));
            }
        }
    }
    float bs[UB][4], wx[UB][4][P > 0 ? P : 1];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bs[ub][g] = bias[g * SF_H + unit0 + ub];
#pragma unroll
            for (int q = 0; q < P; ++q) wx[ub][g][q] = Wx[(long)(g * SF_H + unit0 + ub) * P + q];
        }
    }

    const long ntiles = (M + 15) / 16;
    // Two tiles ahead of the one being multiplied are in flight in registers (s1: tile k+1,
    // s2: tile k+2).  The loop body has no branch (loads past the walker's last tile re-read
    // that tile, FULL drops the row predicates), so the compiler counts outstanding memory
    // operations exactly; with branches it waits for ALL of them -- the stores of the tile
    // just finished included -- before it touches a prefetched tile, once per tile.
    struct Stage {
        f32x4v xr[V4];
        float cp[4][UB], xv[4][P > 0 ? P : 1];
    };
    auto load_tile = [&](long tile, Stage& d) {
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int flat = 4 * (tid + 256 * i), row = flat / KD, col = flat % KD;
            const long grow = tile * 16 + row;
            const bool from_x = P == 0 && col < SF_H;
            const float* src = from_x ? x : h_prev;
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            // (no test of the pointer itself: that would be a branch in the tile loop)
            d.xr[i] = ((FULL || grow < M) && (from_x || STATE))
                          ? *reinterpret_cast<const f32x4v*>(src + grow * SF_H + (col & (SF_H - 1)))
                          : z;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long grow = tile * 16 + 4 * s + r;
            const bool ok = FULL || grow < M;
#pragma unroll
            for (int ub = 0; ub < UB; ++ub)
                d.cp[r][ub] = (ok && STATE) ? c_prev[grow * SF_H + unit0 + ub] : 0.f;
#pragma unroll
            for (int q = 0; q < P; ++q) d.xv[r][q] = ok ? x[grow * P + q] : 0.f;
        }
    };
    auto put_tile = [&](int buf, const Stage& d) {
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int flat = 4 * (tid + 256 * i), row = flat / KD, col = flat % KD;
            u32x2 h, l;
            unsigned a, b;
            sf_split_pair(d.xr[i][0], d.xr[i][1], a, b);
            h[0] = a; l[0] = b;
            sf_split_pair(d.xr[i][2], d.xr[i][3], a, b);
            h[1] = a; l[1] = b;
            *reinterpret_cast<u32x2*>(&xt[buf].hi[row * LD + col]) = h;
            *reinterpret_cast<u32x2*>(&xt[buf].lo[row * LD + col]) = l;
        }
    };

    // tile k of this walker is tile index walker + k * walkers
    const long mine = walker < ntiles ? (ntiles - walker + walkers - 1) / walkers : 0;
    if (mine == 0) return;                       // the whole workgroup
    const long last = walker + (mine - 1) * walkers;
    // (a 32-bit minimum: the 64-bit selection compiles to a branch, which ends the block the
    // scheduler interleaves the tile loop's work in; the host refuses more than 2^31 tiles)
    const int last_k = (int)(mine - 1);
    auto tile_of = [&](long k) { return walker + (long)min((int)k, last_k) * walkers; };
    Stage s0, s1, s2;
    load_tile(tile_of(0), s0);
    load_tile(tile_of(1), s1);
    load_tile(tile_of(2), s2);
    float cnow[4][UB], xnow[4][P > 0 ? P : 1];
    auto take = [&](const Stage& d) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) cnow[r][ub] = d.cp[r][ub];
#pragma unroll
            for (int q = 0; q < P; ++q) xnow[r][q] = d.xv[r][q];
        }
    };
    put_tile(0, s0);
    take(s0);
    __syncthreads();
    // The cell update of a tile (exp / rcp chains, stores) does not depend on the matrix
    // instructions of the NEXT tile, and with one wavefront per SIMD nothing else could fill
    // the matrix core's shadow: iteration k multiplies tile k and, between its groups of
    // matrix instructions, finishes tile k-1 from `accp` / `cprev`.
    f32x4v acc[UB][4], accp[UB][4];
    float cthis[4][UB], cprev[4][UB];
    auto begin_tile = [&]() {
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float a = bs[ub][g];
#pragma unroll
                    for (int q = 0; q < P; ++q) a += xnow[r][q] * wx[ub][g][q];
                    acc[ub][g][r] = a;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) cthis[r][ub] = cnow[r][ub];
        }
    };
    // tile k+1 (in s1 since the previous iteration) into the other LDS buffer, s2 moves up,
    // tile k+3 is requested
    auto advance = [&](long k, int buf) {
        put_tile(buf ^ 1, s1);
        take(s1);
        s1 = s2;
        load_tile(tile_of(k + 3), s2);
    };
    auto retire = [&]() {
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
            for (int g = 0; g < 4; ++g) accp[ub][g] = acc[ub][g];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) cprev[r][ub] = cthis[r][ub];
        }
    };
    // cell update of row 4s + r of `tile` for this lane's unit(s), from accp / cprev
    auto cell_row = [&](int r, long tile) {
        const long grow = tile * 16 + 4 * s + r;
        float hn[UB], cn[UB];
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            const float a_i = sigmoidf_(accp[ub][0][r]), a_f = sigmoidf_(accp[ub][1][r]);
            const float a_g = tanhf_(accp[ub][2][r]), a_o = sigmoidf_(accp[ub][3][r]);
            cn[ub] = a_f * cprev[r][ub] + a_i * a_g;
            hn[ub] = a_o * tanhf_(cn[ub]);
        }
        if (FULL || grow < M) {
            if constexpr (UB == 2) {
                store_streaming(h_out + grow * SF_H + unit0, make_float2(hn[0], hn[1]));
                store_streaming(c_out + grow * SF_H + unit0, make_float2(cn[0], cn[1]));
            } else {
                __builtin_nontemporal_store(hn[0], h_out + grow * SF_H + unit0);
                __builtin_nontemporal_store(cn[0], c_out + grow * SF_H + unit0);
            }
        }
    };
    constexpr int GROUPS = KK / 2, ROWS_PER_GROUP = 4 / GROUPS;
    const int a_off = cc * LD + 8 * s;
    // the matrix instructions of the tile in LDS buffer `buf`, in GROUPS groups; between(grp)
    // is issued with group grp
    auto products = [&](int buf, auto&& between) {
        bf16x8 ah[2][2], al[2][2];
        auto read_group = [&](int grp, bf16x8 (&h_)[2], bf16x8 (&l_)[2]) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                h_[q] = *reinterpret_cast<const bf16x8*>(&xt[buf].hi[a_off + 32 * (2 * grp + q)]);
                l_[q] = *reinterpret_cast<const bf16x8*>(&xt[buf].lo[a_off + 32 * (2 * grp + q)]);
            }
        };
        read_group(0, ah[0], al[0]);
#pragma unroll
        for (int grp = 0; grp < GROUPS; ++grp) {
            if (grp + 1 < GROUPS) read_group(grp + 1, ah[(grp + 1) & 1], al[(grp + 1) & 1]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int kk = 2 * grp + q;
#pragma unroll
                for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[ub][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], whi[ub][g][kk], acc[ub][g], 0, 0, 0);
                }
#pragma unroll
                for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[ub][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[grp & 1][q], whi[ub][g][kk], acc[ub][g], 0, 0, 0);
                }
#pragma unroll
                for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[ub][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], wlo[ub][g][kk], acc[ub][g], 0, 0, 0);
                }
            }
            between(grp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // tile 0: nothing to finish yet
    begin_tile();
    advance(0, 0);
    products(0, [](int) {});
    // (orders the LDS writes of the next tile before its reads; the buffer written in the
    // next iteration was last read in this one, before this barrier)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    retire();
    int buf = 1;
    for (long k = 1; k < mine; ++k) {
        const long done = walker + (k - 1) * walkers;
        begin_tile();
        advance(k, buf);
        products(buf, [&](int grp) {
#pragma unroll
            for (int rr = 0; rr < ROWS_PER_GROUP; ++rr) cell_row(ROWS_PER_GROUP * grp + rr, done);
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        retire();
        buf ^= 1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cell_row(r, last);
}

// The upper layer's instances are compiled apart (KVAE_SF_UPPER, with the compiler option that
// keeps matrix-instruction results in ordinary registers: its accumulators are then read by
// the cell update without copies, 406 -> 373 instructions per tile; the lower layer's
// instances are slower with that option).
int launch_lstm128_step_upper(long B, const float* x, const float* Wx, const float* h_prev,
                              const float* c_prev, const float* WhhT, const float* bias,
                              float* h_out, float* c_out, int cus, hipStream_t s);
// The lower layer's two-input instances, likewise (KVAE_SF_LOWER2).
int launch_lstm128_step_lower2(long B, const float* x, const float* Wx, const float* h_prev,
                               const float* c_prev, const float* WhhT, const float* bias,
                               float* h_out, float* c_out, int cus, hipStream_t s);

#define KVAE_SF_ARGS B, x, Wx, h_prev, c_prev, WhhT, bias, h_out, c_out
#define KVAE_SF_LAUNCH(P_, grid_)                                                             \
    do {                                                                                      \
        if (full && state)                                                                    \
            hipLaunchKernelGGL((lstm128_step_fused_kernel<P_, true, true>), grid_, dim3(256), \
                               0, s, KVAE_SF_ARGS);                                           \
        else if (state)                                                                       \
            hipLaunchKernelGGL((lstm128_step_fused_kernel<P_, false, true>), grid_, dim3(256), \
                               0, s, KVAE_SF_ARGS);                                           \
        else                                                                                  \
            hipLaunchKernelGGL((lstm128_step_fused_kernel<P_, false, false>), grid_, dim3(256), \
                               0, s, KVAE_SF_ARGS);                                           \
    } while (0)

#ifdef KVAE_SF_UPPER
int launch_lstm128_step_upper(long B, const float* x, const float* Wx, const float* h_prev,
                              const float* c_prev, const float* WhhT, const float* bias,
                              float* h_out, float* c_out, int cus, hipStream_t s) {
    const bool full = B % 16 == 0, state = h_prev != nullptr;
    const long tiles = (B + 15) / 16;
    // tile walkers in groups of 8 (one per XCD), two workgroups (unit halves) per walker,
    // one workgroup per compute unit
    long groups = cus / 16;
    if (groups < 1) groups = 1;
    if ((tiles + 7) / 8 < groups) groups = (tiles + 7) / 8;
    const dim3 grid((unsigned)(16 * groups));
    KVAE_SF_LAUNCH(0, grid);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}
#endif

#ifdef KVAE_SF_LOWER2
int launch_lstm128_step_lower2(long B, const float* x, const float* Wx, const float* h_prev,
                               const float* c_prev, const float* WhhT, const float* bias,
                               float* h_out, float* c_out, int cus, hipStream_t s) {
    const bool full = B % 16 == 0, state = h_prev != nullptr;
    const long tiles = (B + 15) / 16;
    const dim3 grid((unsigned)(tiles < cus ? tiles : cus));
    KVAE_SF_LAUNCH(2, grid);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}
#endif

}  // namespace kvae

#if !defined(KVAE_SF_UPPER) && !defined(KVAE_SF_LOWER2)
using namespace kvae;

extern "C" {

int kvae_lstm128_step_fused(long B, int P, const void* x_, const void* Wx_, const void* h_prev_,
                            const void* c_prev_, const void* WhhT_, const void* bias_,
                            void* h_out_, void* c_out_, void* stream) {
    if (B < 1 || B > (1L << 34) || !x_ || !Wx_ || !WhhT_ || !bias_ || !h_out_ || !c_out_ ||
        (h_prev_ == nullptr) != (c_prev_ == nullptr))
        return KVAE_E_BADARG;
    // wider inputs do not leave the registers for the tiles in flight: use kvae_lstm128_forward
    if (P < 0 || P > 2) return KVAE_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(h_prev_) | (P == 0 ? reinterpret_cast<uintptr_t>(x_) : 0)) & 15)
        return KVAE_E_BADARG;
    const int cus = device_compute_units();
    const float *x = (const float*)x_, *Wx = (const float*)Wx_, *h_prev = (const float*)h_prev_,
                *c_prev = (const float*)c_prev_, *WhhT = (const float*)WhhT_,
                *bias = (const float*)bias_;
    float *h_out = (float*)h_out_, *c_out = (float*)c_out_;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0)
        return launch_lstm128_step_upper(B, x, Wx, h_prev, c_prev, WhhT, bias, h_out, c_out, cus, s);
    const bool full = B % 16 == 0, state = h_prev != nullptr;
    const long tiles = (B + 15) / 16;
    const dim3 grid((unsigned)(tiles < cus ? tiles : cus));
    if (P == 2)
        return launch_lstm128_step_lower2(B, x, Wx, h_prev, c_prev, WhhT, bias, h_out, c_out, cus, s);
    KVAE_SF_LAUNCH(1, grid);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
#endif
