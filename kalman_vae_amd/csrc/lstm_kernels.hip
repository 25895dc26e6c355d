// gfx950 kernels for the recurrent part of the LSTM dynamics-parameter network
// (the producer of the mixing weights alpha; reference:
// kvae/dynamics_parameter_network.py:46-58, torch.nn.LSTM semantics, gate order
// i, f, g, o, zero initial state).
//
// Split of the work for one layer over a whole sequence:
//   * input projection for ALL time steps, Gx = X W_ih^T + b_ih + b_hh: one
//     large time-batched GEMM -- a plain library GEMM, done by the caller;
//   * the recurrence, T dependent steps: THIS FILE.  Per step one launch of
//     lstm_step_forward_kernel: gates = Gx[t] + h_{t-1} W_hh^T on the matrix
//     cores (exact fp32, v_mfma_f32_32x32x2_f32) fused with the cell update, so
//     the pre-activations never leave registers;
//   * backward per step: lstm_cell_backward_kernel (element-wise gate adjoints)
//     and gemm_acc_kernel (dh_{t-1} += dG_t W_hh on the matrix cores);
//   * weight gradients dW = dG^T [X | H_prev]: large-K GEMMs over all steps at
//     once, again plain library GEMMs done by the caller.
//
// Tiling: a wavefront owns 32 sequences x 32 hidden units and keeps the four
// gate pre-activations of that tile in four 32x32 fp32 accumulators (64 VGPRs);
// lane l holds column (unit) l & 31 and 16 of the 32 rows, the same rows in all
// four accumulators, so the cell update is purely per-lane.  The 32 x H slab of
// h_{t-1} that feeds the A operand is staged once in LDS (row stride H + 1:
// the per-lane reads walk a column); the B operand W_hh^T[k][gate*H + unit] is
// read straight from L2 (lanes = consecutive units, coalesced).
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"

namespace kvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// row of accumulator register r for lane-half hh (C/D layout of 32x32 MFMA)
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

constexpr int LSTM_MAX_H = 256;

// Copy a 32 x COLS tile (rows `src_ld` apart in global memory, rows >= rows_valid
// read as zero) into LDS with row stride COLS + 1.  All of a lane's 16-byte loads
// are issued before the first LDS write, so the tile costs one memory latency,
// not one per element.
template <int COLS>
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, long src_ld,
                                           int rows_valid, float* a_tile) {
    constexpr int F4_PER_ROW = COLS / 4, PER_LANE = 32 * F4_PER_ROW / 64, lda = COLS + 1;
    const int lane = threadIdx.x;
    float4 v[PER_LANE];
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) {
        const int f = lane + 64 * q, row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
        v[q] = row < rows_valid
                   ? *reinterpret_cast<const float4*>(src + (long)row * src_ld + 4 * c4)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < PER_LANE; ++q) {
        const int f = lane + 64 * q, row = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
        float* d = a_tile + row * lda + 4 * c4;
        d[0] = v[q].x;
        d[1] = v[q].y;
        d[2] = v[q].z;
        d[3] = v[q].w;
    }
}
constexpr int KCH = 8;  // k-pairs whose B operands are fetched together (latency hiding)

// grid = (ceil(B/32), H/32), block = 64.
// gates: (B, 4H) row-major; in: Gx[t] (input projection + biases), out: activated
// gates i, f, g, o (kept for the backward).  h_prev/c_prev: (B,H) or null (= 0).
// The B operands (rows of W_hh^T, L2-resident) are fetched KCH k-pairs ahead
// into registers, double-buffered, so that their latency overlaps the MFMAs.
template <int H>
__global__ void __launch_bounds__(64)
lstm_step_forward_kernel(int B, float* __restrict__ gates,
                         const float* __restrict__ WhhT,  // (H, 4H)
                         const float* __restrict__ h_prev, const float* __restrict__ c_prev,
                         float* __restrict__ h_out, float* __restrict__ c_out) {
    constexpr int lda = H + 1, G4 = 4 * H, NKK = H / 2;
    __shared__ float a_tile[32 * lda];
    const int lane = threadIdx.x, j = lane & 31, hh = lane >> 5;
    const int r0 = blockIdx.x * 32, u0 = blockIdx.y * 32;

    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + acc_row(r, hh);
            acc[g][r] = row < B ? gates[(long)row * G4 + g * H + u0 + j] : 0.f;
        }
    }
    if (h_prev) {
        const float* wb = WhhT + u0 + j + (long)hh * G4;  // row 2kk+hh, column u0+j
        float bcur[KCH][4], bnxt[KCH][4];
#pragma unroll
        for (int u = 0; u < KCH; ++u) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bcur[u][g] = wb[(long)(2 * u) * G4 + g * H];
        }
        // stage h_{t-1}[r0 : r0+32][0 : H] in LDS
        stage_tile<H>(h_prev + (long)r0 * H, H, B - r0, a_tile);
        __syncthreads();
        for (int kk0 = 0; kk0 < NKK; kk0 += KCH) {
            if (kk0 + KCH < NKK) {
#pragma unroll
                for (int u = 0; u < KCH; ++u) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        bnxt[u][g] = wb[(long)(2 * (kk0 + KCH + u)) * G4 + g * H];
                }
            }
#pragma unroll
            for (int u = 0; u < KCH; ++u) {
                const float a = a_tile[j * lda + 2 * (kk0 + u) + hh];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[u][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[u][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[u][2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[u][3], acc[3], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < KCH; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bcur[u][g] = bnxt[u][g];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + acc_row(r, hh);
        if (row < B) {
            const long o = (long)row * H + u0 + j;
            const float ig = sigmoidf_(acc[0][r]);
            const float fg = sigmoidf_(acc[1][r]);
            const float gg = tanhf(acc[2][r]);
            const float og = sigmoidf_(acc[3][r]);
            const float cp = c_prev ? c_prev[o] : 0.f;
            const float c = fg * cp + ig * gg;
            h_out[o] = og * tanhf(c);
            c_out[o] = c;
            float* gr = gates + (long)row * G4 + u0 + j;
            gr[0] = ig;
            gr[H] = fg;
            gr[2 * H] = gg;
            gr[3 * H] = og;
        }
    }
}

// One thread per (sequence, unit).  dh: gradient w.r.t. h_t (upstream + recurrent),
// dc: carried gradient w.r.t. c_t (in: from step t+1, out: for step t-1).
__global__ void __launch_bounds__(256)
lstm_cell_backward_kernel(long total, int H, const float* __restrict__ gates,
                          const float* __restrict__ c_prev, const float* __restrict__ c_t,
                          const float* __restrict__ dh, float* __restrict__ dc,
                          float* __restrict__ dgates, int first_in_time_reversed) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long row = idx / H;
    const int u = (int)(idx % H);
    const float* gr = gates + row * 4 * H + u;
    const float ig = gr[0], fg = gr[H], gg = gr[2 * H], og = gr[3 * H];
    const float tc = tanhf(c_t[idx]);
    const float dht = dh[idx];
    const float dcn = first_in_time_reversed ? 0.f : dc[idx];
    const float dct = dht * og * (1.f - tc * tc) + dcn;
    const float cp = c_prev ? c_prev[idx] : 0.f;
    float* dg = dgates + row * 4 * H + u;
    dg[0] = dct * gg * ig * (1.f - ig);
    dg[H] = dct * cp * fg * (1.f - fg);
    dg[2 * H] = dct * ig * (1.f - gg * gg);
    dg[3 * H] = dht * tc * og * (1.f - og);
    dc[idx] = dct * fg;
}

// C (M x N) += A (M x Kd) * Bm (Kd x N), all row-major fp32; N a multiple of 32,
// Kd a multiple of 128.  grid = (ceil(M/32), N/32), block = 64.  A is staged
// through LDS in chunks of 128 columns; B rows are prefetched GCH k-pairs ahead.
constexpr int GCH = 16;
__global__ void __launch_bounds__(64)
gemm_acc_kernel(int M, int N, int Kd, const float* __restrict__ A, const float* __restrict__ Bm,
                float* __restrict__ C) {
    constexpr int KC = 128, lda = KC + 1;
    __shared__ float a_tile[32 * lda];
    const int lane = threadIdx.x, j = lane & 31, hh = lane >> 5;
    const int r0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + acc_row(r, hh);
        acc[r] = row < M ? C[(long)row * N + n0 + j] : 0.f;
    }
    const float* bb = Bm + n0 + j + (long)hh * N;  // row 2kk+hh, column n0+j
    const int nkk = Kd / 2;
    float bcur[GCH], bnxt[GCH];
#pragma unroll
    for (int u = 0; u < GCH; ++u) bcur[u] = bb[(long)(2 * u) * N];
    for (int k0 = 0; k0 < Kd; k0 += KC) {
        __syncthreads();
        stage_tile<KC>(A + (long)r0 * Kd + k0, Kd, M - r0, a_tile);
        __syncthreads();
        for (int kk0 = 0; kk0 < KC / 2; kk0 += GCH) {
            const int kkg = k0 / 2 + kk0;  // global k-pair index of this chunk
            if (kkg + GCH < nkk) {
#pragma unroll
                for (int u = 0; u < GCH; ++u) bnxt[u] = bb[(long)(2 * (kkg + GCH + u)) * N];
            }
#pragma unroll
            for (int u = 0; u < GCH; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_tile[j * lda + 2 * (kk0 + u) + hh],
                                                          bcur[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < GCH; ++u) bcur[u] = bnxt[u];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + acc_row(r, hh);
        if (row < M) C[(long)row * N + n0 + j] = acc[r];
    }
}

}  // namespace kvae

using namespace kvae;

extern "C" {

int kvae_lstm_layer_forward(int T, int B, int H, void* gates, const void* WhhT, void* h, void* c,
                            void* stream) {
    if (H != 32 && H != 64 && H != 128 && H != 256) return KVAE_E_UNSUPPORTED;
    if (T < 1 || B < 1 || !gates || !WhhT || !h || !c) return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((B + 31) / 32, H / 32);
    const long BH = (long)B * H, BG = (long)B * 4 * H;
    float* g = (float*)gates;
    float* hp = (float*)h;
    float* cp = (float*)c;
    for (int t = 0; t < T; ++t) {
        const float* hprev = t ? hp + (t - 1) * BH : nullptr;
        const float* cprev = t ? cp + (t - 1) * BH : nullptr;
#define KVAE_LSTM_LAUNCH(H_)                                                                     \
    hipLaunchKernelGGL(lstm_step_forward_kernel<H_>, grid, dim3(64), 0, s, B, g + t * BG,       \
                       (const float*)WhhT, hprev, cprev, hp + t * BH, cp + t * BH)
        switch (H) {
            case 32: KVAE_LSTM_LAUNCH(32); break;
            case 64: KVAE_LSTM_LAUNCH(64); break;
            case 128: KVAE_LSTM_LAUNCH(128); break;
            case 256: KVAE_LSTM_LAUNCH(256); break;
            default: return KVAE_E_UNSUPPORTED;
        }
#undef KVAE_LSTM_LAUNCH
    }
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_lstm_layer_backward(int T, int B, int H, const void* gates, const void* c, void* dh,
                             const void* Whh, void* dgates, void* dc_work, void* stream) {
    if (H != 32 && H != 64 && H != 128 && H != 256) return KVAE_E_UNSUPPORTED;
    if (T < 1 || B < 1 || !gates || !c || !dh || !Whh || !dgates || !dc_work) return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const long BH = (long)B * H, BG = (long)B * 4 * H;
    const float* g = (const float*)gates;
    const float* cc = (const float*)c;
    float* dhp = (float*)dh;
    float* dg = (float*)dgates;
    const dim3 ggrid((B + 31) / 32, H / 32);
    for (int t = T - 1; t >= 0; --t) {
        hipLaunchKernelGGL(lstm_cell_backward_kernel, dim3((BH + 255) / 256), dim3(256), 0, s, BH, H,
                           g + t * BG, t ? cc + (t - 1) * BH : nullptr, cc + t * BH, dhp + t * BH,
                           (float*)dc_work, dg + t * BG, t == T - 1 ? 1 : 0);
        if (t > 0) {
            // dh_{t-1} += dG_t W_hh   (B x 4H) (4H x H)
            hipLaunchKernelGGL(gemm_acc_kernel, ggrid, dim3(64), 0, s, B, H, 4 * H, dg + t * BG,
                               (const float*)Whh, dhp + (t - 1) * BH);
        }
    }
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
