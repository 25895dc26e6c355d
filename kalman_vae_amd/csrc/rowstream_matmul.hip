// Y = X W (+ bias) for a tall X (M rows, M ~ 10^5) and a small W (K x N, K N = 65536):
// the two time-batched products of the dynamics-parameter network's LSTM that are not
// part of a recurrence,
//   layer 1's input projection of all steps      (T B, 128) x (128, 512) + bias
//   the gradient handed from layer 1 to layer 0  (T B, 512) x (512, 128)
// (torch.nn.LSTM inside kvae/dynamics_parameter_network.py:46-58 and its autograd).
//
// Both are bound by streaming X and Y through HBM once (W is 256 KB).  A workgroup of four
// wavefronts (one per SIMD, 512 registers per lane) keeps all of W in registers as the B
// operands of v_mfma_f32_16x16x32_bf16, split into a hi and a lo bf16 half, and works
// through 16-row tiles of X: global -> registers -> split -> LDS (double buffered, the next
// tile's loads in flight while this tile's matrix instructions run) -> A operands.  Every
// fp32 product is a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32 accumulation (about 1e-5
// relative per product, 5e-5 at worst; see lstm_resident.hip).  Wavefront w owns N/4 adjacent output
// columns, lane (cc, s) of it rows 4s..4s+3 and the N/64 adjacent columns from
// w N/4 + cc N/64, so Y is written in 32-byte (N = 512) or 8-byte (N = 128) pieces.
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "lstm_common.h"

namespace kvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (x, y) -> packed bf16 pairs hi = (bf16 x, bf16 y), lo = (bf16 (x - hi.x), bf16 (y - hi.y))
__device__ __forceinline__ void split_pair(float x, float y, unsigned& hi, unsigned& lo) {
    const f32x2v v = {x, y};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2v r = {x - __builtin_bit_cast(float, hi << 16),
                      y - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(x[2 * j], x[2 * j + 1], h[j], l[j]);
    const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
    hi = __builtin_bit_cast(bf16x8, hv);
    lo = __builtin_bit_cast(bf16x8, lv);
}

__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// grid = min(tiles, compute units), block = 256.  X (M,KD), W (KD,ND), bias (ND) or null,
// Y (M,ND), all fp32 row-major.
// FULL: M is a multiple of 16 (no row predicates).
template <int KD, int ND, bool FULL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
rowstream_matmul_kernel(long M, const float* __restrict__ X, const float* __restrict__ W,
                        const float* __restrict__ bias, float* __restrict__ Y) {
    constexpr int NB = ND / 64;                 // 16-column blocks per wavefront
    constexpr int KK = KD / 32;                 // 32-deep k-groups
    constexpr int LD = KD + 8;                  // bf16 per LDS row: rows 4 banks apart
    constexpr int V4 = KD * 16 / (256 * 4);     // float4 loads per lane per tile
    constexpr int GRP = KK < 4 ? KK : 2;        // k-groups whose A operands are read together
    static_assert(KD * ND == 65536 && KD % 32 == 0 && ND % 64 == 0 && V4 >= 1, "shape");
    struct Tile {
        __bf16 hi[16 * LD], lo[16 * LD];
    };
    __shared__ __attribute__((aligned(16))) Tile xt[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cc = lane & 15, s = lane >> 4;
    const int cbase = wave * (ND / 4) + NB * cc;

    // w*[j][kk] multiplies X[., 32 kk + 8 s + .] into output column cbase + j
    bf16x8 whi[NB][KK], wlo[NB][KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        float v[NB][8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float* wr = W + (long)(32 * kk + 8 * s + jj) * ND + cbase;
            if constexpr (NB == 8) {
                const float4 a = *reinterpret_cast<const float4*>(wr);
                const float4 b = *reinterpret_cast<const float4*>(wr + 4);
                v[0][jj] = a.x; v[1][jj] = a.y; v[2][jj] = a.z; v[3][jj] = a.w;
                v[4][jj] = b.x; v[5][jj] = b.y; v[6][jj] = b.z; v[7][jj] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) v[j][jj] = wr[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) split8(v[j], whi[j][kk], wlo[j][kk]);
    }
    float bs[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bs[j] = bias ? bias[cbase + j] : 0.f;

    const long ntiles = (M + 15) / 16;
    // Two tiles ahead of the one being multiplied are in flight in registers.  The loop body
    // has no branch (loads past the workgroup's last tile re-read that tile, FULL drops the row
    // predicates), so the compiler counts outstanding memory operations exactly; with branches
    // it waits for ALL of them -- the stores of the tile just finished included -- before it
    // touches a prefetched tile, once per tile.
    struct Stage {
        f32x4v xr[V4];
    };
    auto load_tile = [&](long tile, Stage& d) {
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int flat = 4 * (tid + 256 * i), row = flat / KD, col = flat % KD;
            const long grow = tile * 16 + row;
            const f32x4v z = {0.f, 0.f, 0.f, 0.f};
            d.xr[i] = (FULL || grow < M) ? __builtin_nontemporal_load(
                                               reinterpret_cast<const f32x4v*>(X + grow * KD + col))
                                         : z;
        }
    };
    auto put_tile = [&](int buf, const Stage& d) {
#pragma unroll
        for (int i = 0; i < V4; ++i) {
            const int flat = 4 * (tid + 256 * i), row = flat / KD, col = flat % KD;
            u32x2 h, l;
            unsigned a, b;
            split_pair(d.xr[i][0], d.xr[i][1], a, b);
            h[0] = a; l[0] = b;
            split_pair(d.xr[i][2], d.xr[i][3], a, b);
            h[1] = a; l[1] = b;
            *reinterpret_cast<u32x2*>(&xt[buf].hi[row * LD + col]) = h;
            *reinterpret_cast<u32x2*>(&xt[buf].lo[row * LD + col]) = l;
        }
    };

    // tile k of this workgroup is tile index blockIdx.x + k * gridDim.x
    const long first = blockIdx.x, stride = gridDim.x;
    const long mine = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
    if (mine == 0) return;                       // the whole workgroup
    const long last = first + (mine - 1) * stride;
    auto tile_of = [&](long k) { return k < mine ? first + k * stride : last; };
    // s1, s2: the tiles after the one in LDS, in flight in registers; they take the role of
    // `next` in turn (a copy between them would wait for the younger one's loads), so the loop
    // is unrolled by two and an odd last tile is peeled
    Stage s1, s2;
    {
        Stage s0;
        load_tile(tile_of(0), s0);
        load_tile(tile_of(1), s1);
        load_tile(tile_of(2), s2);
        put_tile(0, s0);
    }
    __syncthreads();
    auto one_tile = [&](long k, int buf, Stage& next) {
        const long tile = first + k * stride;
        f32x4v acc[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        const int a_off = cc * LD + 8 * s;
        bf16x8 ah[2][GRP], al[2][GRP];
        auto read_group = [&](int grp, bf16x8 (&h_)[GRP], bf16x8 (&l_)[GRP]) {
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                h_[q] = *reinterpret_cast<const bf16x8*>(&xt[buf].hi[a_off + 32 * (GRP * grp + q)]);
                l_[q] = *reinterpret_cast<const bf16x8*>(&xt[buf].lo[a_off + 32 * (GRP * grp + q)]);
            }
        };
        read_group(0, ah[0], al[0]);
        // tile k+1 into the other LDS buffer (after the last tile: a copy of it that nothing
        // reads), then tile k+3 is requested into the same registers
        put_tile(buf ^ 1, next);
        load_tile(tile_of(k + 3), next);
#pragma unroll
        for (int grp = 0; grp < KK / GRP; ++grp) {
            if (grp + 1 < KK / GRP) read_group(grp + 1, ah[(grp + 1) & 1], al[(grp + 1) & 1]);
#pragma unroll
            for (int q = 0; q < GRP; ++q) {
                const int kk = GRP * grp + q;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], whi[j][kk], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[grp & 1][q], whi[j][kk], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[grp & 1][q], wlo[j][kk], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // rows 4s..4s+3 of the tile, columns cbase .. cbase + NB - 1
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long grow = tile * 16 + 4 * s + r;
            if (FULL || grow < M) {
                float* yr = Y + grow * ND + cbase;
                if constexpr (NB == 8) {
                    const f32x4v a = {acc[0][r] + bs[0], acc[1][r] + bs[1], acc[2][r] + bs[2],
                                      acc[3][r] + bs[3]};
                    const f32x4v b = {acc[4][r] + bs[4], acc[5][r] + bs[5], acc[6][r] + bs[6],
                                      acc[7][r] + bs[7]};
                    __builtin_nontemporal_store(a, reinterpret_cast<f32x4v*>(yr));
                    __builtin_nontemporal_store(b, reinterpret_cast<f32x4v*>(yr + 4));
                } else {
                    static_assert(NB == 8 || NB == 2, "store width");
                    const f32x2v a = {acc[0][r] + bs[0], acc[1][r] + bs[1]};
                    __builtin_nontemporal_store(a, reinterpret_cast<f32x2v*>(yr));
                }
            }
        }
        // orders the writes of the next tile before its reads; the buffer written in the next
        // call was last read in this one, before this barrier
        lds_only_barrier();
    };
    long k = 0;
    for (; k + 1 < mine; k += 2) {
        one_tile(k, 0, s1);
        one_tile(k + 1, 1, s2);
    }
    if (k < mine) one_tile(k, 0, s1);
}

}  // namespace kvae

using namespace kvae;

extern "C" {

int kvae_rowstream_matmul(long M, int K, int N, const void* X, const void* W, const void* bias,
                          void* Y, void* stream) {
    if (M < 1 || !X || !W || !Y) return KVAE_E_BADARG;
    if (!((K == 128 && N == 512) || (K == 512 && N == 128))) return KVAE_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W) |
         reinterpret_cast<uintptr_t>(Y)) & 15)
        return KVAE_E_BADARG;
    const int cus = device_compute_units();
    const long tiles = (M + 15) / 16;
    const dim3 grid((unsigned)(tiles < cus ? tiles : cus));
#define KVAE_RS_LAUNCH(K_, N_, F_)                                                          \
    hipLaunchKernelGGL((rowstream_matmul_kernel<K_, N_, F_>), grid, dim3(256), 0,            \
                       (hipStream_t)stream, M, (const float*)X, (const float*)W,             \
                       (const float*)bias, (float*)Y)
    const bool full = M % 16 == 0;
    if (K == 128) {
        if (full) KVAE_RS_LAUNCH(128, 512, true); else KVAE_RS_LAUNCH(128, 512, false);
    } else {
        if (full) KVAE_RS_LAUNCH(512, 128, true); else KVAE_RS_LAUNCH(512, 128, false);
    }
#undef KVAE_RS_LAUNCH
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
