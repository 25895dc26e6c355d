// Weight gradient of one LSTM layer of the dynamics-parameter network (hidden size 128):
//     dW = dG^T H,   dG (R, 512) gate gradients, H (R, 128) the layer's input or its own
//     previous h, R = T B rows (~10^5)
// (autograd of torch.nn.LSTM inside kvae/dynamics_parameter_network.py:46-58).  H may be
// paired with a row offset (h_{t-1}: H[m - B]).
//
// The product is bound by reading dG and H once (210 MB at R = 81920; the fp32 library
// GEMM is bound by the fp32 matrix instruction instead).  The R rows are cut into slabs; a
// workgroup owns one slab and a 128-column slice of dG, i.e. a 128 x 128 piece of the
// output.  The matrix instruction (v_mfma_f32_16x16x32_bf16) contracts over 32 rows and
// wants, per lane, 8 consecutive rows of ONE column of each operand.  A lane loads 8 rows x
// 4 adjacent columns from global memory (8 float4 loads; a wavefront's load instruction
// covers whole 512-byte row pieces), which is four such operand pieces; it splits each
// into bf16 hi and lo halves and writes them to LDS as 16-byte pieces, so the
// transposition costs nothing.  Products are a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32
// accumulation (about 1e-5 relative per product, 5e-5 at worst).
//
// A workgroup is eight wavefronts, two per SIMD, and a step is 64 rows: every lane loads
// one (8-row, 4-column) item of it, wavefronts 0-3 multiply rows 0-31 and wavefronts 4-7
// rows 32-63, each into its own 64 x 64 accumulators (a quarter of the output piece), so
// the split / LDS traffic of one wavefront overlaps the matrix instructions of the other
// on its SIMD.  Loads run two steps ahead (a register ring).  At the end the two halves
// are added through LDS.  The 4 workgroups that share a slab (and therefore H) are placed
// on one XCD so that H comes from that XCD's L2 after its first read.  Per-slab results go
// to `partial`; the caller sums the slabs in a fixed order (no atomics).
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "lstm_common.h"

namespace kvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wg_split_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const f32x2v v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2v r = {x - __builtin_bit_cast(float, hi << 16),
                      y - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void wg_split8(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, l;
        wg_split_pair(x[2 * j], x[2 * j + 1], h, l);
        hi[j] = h;
        lo[j] = l;
    }
}

constexpr int WG_ROWS = 64;     // rows of a step
constexpr int WG_PIECE = 8;     // bf16 of one 16-byte piece: 8 rows of one column
constexpr int WG_GC = 512;      // columns of dG
constexpr int WG_H = 128;       // columns of H
constexpr int WG_AC = 128;      // columns of dG per workgroup
constexpr int WG_TC = WG_AC + WG_H;   // columns of the combined tile: [dG slice | H]

// grid = 8 * ceil(slabs/8) * 4, block = 512.  rows_per_slab is a multiple of 64.
// Row m of dG is paired with row m of H (the launcher applies a row offset to the
// pointers).  Slab s writes its (512, 128) result at partial + s * partial_stride.
// FULL: slabs * rows_per_slab == R, no row exists only in part.  The step loop has no
// branch then (loads past the slab's last step re-read that step), so the compiler counts
// the outstanding loads exactly and a step waits only for its own tile, not for the two
// steps requested behind it.
template <bool FULL>
__global__ void __launch_bounds__(512)
wgrad_tn_kernel(long R, long rows_per_slab, int slabs, const float* __restrict__ dG,
                const float* __restrict__ Hm, float* __restrict__ partial,
                long partial_stride) {
    // A 64-row step of the combined tile as eight planes (one per 8-row group) of 256 16-byte
    // pieces (one per column).  ds_read_b128 serves a wavefront in four groups of 16 lanes
    // that are NOT contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32):
    // a group holds all 16 values of lane & 15 but two values of lane >> 4, i.e. 16 adjacent
    // columns of two planes.  A plane is a multiple of the 256-byte bank row, so the plane does
    // not matter and 16 adjacent columns are conflict-free as long as their pieces fall into
    // 16 different 16-byte slots of the row.  ds_write_b128 serves 8 contiguous lanes at a
    // time, against a 128-byte row: a lane writes column 4 q + e, so 8 lanes are 32 columns
    // apart in steps of 4.  `slot()` permutes the low four column bits (c2, c3, c0 ^ c4, c1)
    // so that both hold.  (A column-major tile with 144-byte columns, conflict-free for
    // contiguous 16-lane groups, measured 0.49 conflict cycles per LDS cycle.)
    struct Tile {
        __bf16 hi[WG_ROWS / 8 * WG_TC * WG_PIECE], lo[WG_ROWS / 8 * WG_TC * WG_PIECE];
    };
    auto slot = [](int col) {
        return (col & ~15) | ((col >> 2) & 3) | ((((col ^ (col >> 4)) & 1)) << 2) | ((col & 2) << 2);
    };
    __shared__ __attribute__((aligned(16))) Tile tl[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cc = lane & 15, s = lane >> 4;
    // the 4 workgroups of a slab on one XCD (consecutive workgroup ids go round the 8 XCDs)
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int per_xcd = (slabs + 7) >> 3;
    const int slab = xcd * per_xcd + (idx >> 2), ct = idx & 3;
    if (slab >= slabs || (idx >> 2) >= per_xcd) return;   // the whole workgroup leaves
    const long m_begin = slab * rows_per_slab;
    const int nks = (int)(rows_per_slab / WG_ROWS);
    const int c0 = WG_AC * ct;
    const int kh = wave >> 2, wm = wave & 1, wn = (wave >> 1) & 1;

    // loader role, fixed over the steps: 8-row group g = wave of the 64-row step, 4-column
    // group q = lane of the combined tile; q < 32: dG, else H
    const int g = wave, q = lane;
    const bool is_a = q < 32;
    const float* src = is_a ? dG + c0 + 4 * q : Hm + 4 * (q - 32);
    const long stride = is_a ? WG_GC : WG_H;
    struct Stage {
        f32x4v v[8];
    };
    auto load_step = [&](int k, Stage& dst) {
        const long m0 = m_begin + (long)WG_ROWS * k + 8 * g;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const long m = m0 + jj;
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            dst.v[jj] = (FULL || m < R) ? *reinterpret_cast<const f32x4v*>(src + m * stride) : z;
        }
    };
    auto put_step = [&](int buf, const Stage& st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) x[jj] = st.v[jj][e];
            u32x4 hi, lo;
            wg_split8(x, hi, lo);
            const int at = (g * WG_TC + slot(4 * q + e)) * WG_PIECE;
            *reinterpret_cast<u32x4*>(&tl[buf].hi[at]) = hi;
            *reinterpret_cast<u32x4*>(&tl[buf].lo[at]) = lo;
        }
    };

    f32x4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    auto step_of = [&](int k) { return k < nks ? k : nks - 1; };
    // s1, s2: the steps after the one whose tile is in LDS, in flight in registers
    Stage s1, s2;
    {
        Stage s0;
        load_step(0, s0);
        load_step(step_of(1), s1);
        load_step(step_of(2), s2);
        put_step(0, s0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // One step: multiply the tile in LDS buffer `buf`; meanwhile `next` (step k+1, in flight
    // since two steps ago) goes into the other buffer and is refilled with step k+3.  The two
    // stages take the role of `next` in turn (a copy between them would wait for the younger
    // one's loads), so the loop below is unrolled by two and an odd last step is peeled.
    auto step = [&](int k, int buf, Stage& next) {
        // this wavefront's operands: 8-row groups 4 kh + s of the step
        bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int blk = 4 * wm + qq;
            const int off = ((4 * kh + s) * WG_TC + slot(16 * blk + cc)) * WG_PIECE;
            ah[qq] = *reinterpret_cast<const bf16x8*>(&tl[buf].hi[off]);
            al[qq] = *reinterpret_cast<const bf16x8*>(&tl[buf].lo[off]);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int blk = WG_AC / 16 + 4 * wn + qq;
            const int off = ((4 * kh + s) * WG_TC + slot(16 * blk + cc)) * WG_PIECE;
            bh[qq] = *reinterpret_cast<const bf16x8*>(&tl[buf].hi[off]);
            bl[qq] = *reinterpret_cast<const bf16x8*>(&tl[buf].lo[off]);
        }
        // (after the slab's last step this writes a copy of it that nothing reads)
        put_step(buf ^ 1, next);
        load_step(step_of(k + 3), next);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        // orders the writes of step k+1 before its reads; the buffer written in the next
        // step is the one read in this one, before this barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    int k = 0;
    for (; k + 1 < nks; k += 2) {
        step(k, 0, s1);
        step(k + 1, 1, s2);
    }
    if (k < nks) step(k, 0, s1);
    // the two row halves: wavefronts 4-7 hand their accumulators over through LDS (the tiles
    // are free after the last barrier above), wavefronts 0-3 add and write
    float* xch = reinterpret_cast<float*>(&tl[0]);      // 4 wavefronts x 64 values x 64 lanes
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xch[(((wave & 3) * 64 + (i * 16 + j * 4 + r)) * 64) + lane] = acc[i][j][r];
            }
        }
    }
    __syncthreads();
    if (kh == 1) return;
    // acc[i][j][r] = output row (dG column) c0 + 16 (4 wm + i) + 4 s + r, column
    // 16 (4 wn + j) + cc
    float* out = partial + (long)slab * partial_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int unit = 16 * (4 * wn + j) + cc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = c0 + 16 * (4 * wm + i) + 4 * s + r;
                out[(long)row * WG_H + unit] =
                    acc[i][j][r] + xch[((wave * 64 + (i * 16 + j * 4 + r)) * 64) + lane];
            }
        }
    }
}

}  // namespace kvae

using namespace kvae;

extern "C" {

int kvae_lstm128_wgrad_slabs(long R) {
    if (R < 1) return 0;
    // up to 64 slabs (256 workgroups) of at least one 64-row step each: a small batch (640
    // rows at 32 sequences of 20 steps) is ten one-step slabs, not one slab of ten steps
    long rps = (R + 63) / 64;
    rps = (rps + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    return (int)((R + rps - 1) / rps);
}

int kvae_lstm128_wgrad(long R, const void* dG, const void* H, long shift, void* partial,
                       long partial_stride, void* stream) {
    if (R < 1 || !dG || !H || !partial || shift < 0 || partial_stride < (long)WG_GC * WG_H)
        return KVAE_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(dG) | reinterpret_cast<uintptr_t>(H)) & 15)
        return KVAE_E_BADARG;
    const int slabs = kvae_lstm128_wgrad_slabs(R);
    // rows m >= shift of dG meet rows m - shift of H: the same product without an offset on
    // the R - shift rows that start at dG + shift * 512 (no such rows: every slab is zero)
    const long rows = R > shift ? R - shift : 0;
    const float* a = (const float*)dG + shift * WG_GC;
    long rps = (rows + slabs - 1) / slabs;
    rps = (rps + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    if (rps < WG_ROWS) rps = WG_ROWS;
    const int per_xcd = (slabs + 7) / 8;
    const dim3 grid(8 * per_xcd * 4);        // (xcd, slab within xcd, column slice)
    if (rps * slabs == rows)
        hipLaunchKernelGGL(wgrad_tn_kernel<true>, grid, dim3(512), 0, (hipStream_t)stream, rows,
                           rps, slabs, a, (const float*)H, (float*)partial, partial_stride);
    else
        hipLaunchKernelGGL(wgrad_tn_kernel<false>, grid, dim3(512), 0, (hipStream_t)stream, rows,
                           rps, slabs, a, (const float*)H, (float*)partial, partial_stride);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
