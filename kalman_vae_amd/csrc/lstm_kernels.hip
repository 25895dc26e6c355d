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
#include <cstdlib>

#include "../../include/kvae_kalman.h"

#include "lstm_common.h"

namespace kvae {

// row of accumulator register r for lane-half hh (C/D layout of 32x32 MFMA)
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

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
#pragma unroll 1
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
            const float gg = tanhf_(acc[2][r]);
            const float og = sigmoidf_(acc[3][r]);
            const float cp = c_prev ? c_prev[o] : 0.f;
            const float c = fg * cp + ig * gg;
            h_out[o] = og * tanhf_(c);
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
    const float tc = tanhf_(c_t[idx]);
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
#pragma unroll 1
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

// ===========================================================================
// Persistent two-layer variant (hidden size 128, the reference's default
// weight network): ONE launch runs all T steps of both layers for a tile of 32
// sequences.  A workgroup has four wavefronts; wavefront w owns hidden units
// [32w, 32w+32) of both layers.  h_t of each layer lives in LDS (double
// buffered over t), c_t in registers; the weights (3 x 256 KiB) are streamed
// from L2 every step with register prefetch.  No per-step launch, no global
// round trip of h between steps; gates/h/c are written out only because the
// backward needs them.
// ===========================================================================
namespace kvae {

constexpr int PH = 128;         // hidden size of the persistent variant
constexpr int PLDA = PH + 1;    // LDS row stride of an h tile
constexpr int PG4 = 4 * PH;

// acc[g] += A_tile (32 x 128, LDS) * Bt[:, g*PH + u0 + j], Bt = (128, 512) row-major
__device__ __forceinline__ void mfma_k128_4gates(f32x16 (&acc)[4], const float* a_tile,
                                                 const float* __restrict__ Bt, int u0) {
    const int lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const float* wb = Bt + u0 + j + (long)hh * PG4;
    float bcur[KCH][4], bnxt[KCH][4];
#pragma unroll
    for (int u = 0; u < KCH; ++u) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bcur[u][g] = wb[(long)(2 * u) * PG4 + g * PH];
    }
#pragma unroll 1
    for (int kk0 = 0; kk0 < PH / 2; kk0 += KCH) {
        if (kk0 + KCH < PH / 2) {
#pragma unroll
            for (int u = 0; u < KCH; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bnxt[u][g] = wb[(long)(2 * (kk0 + KCH + u)) * PG4 + g * PH];
            }
        }
#pragma unroll
        for (int u = 0; u < KCH; ++u) {
            const float a = a_tile[j * PLDA + 2 * (kk0 + u) + hh];
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

// gate activations + cell update for the 16 rows of this lane; writes gates/h/c of
// step t to global memory and h into the LDS tile for the consumers of h_t.
__device__ __forceinline__ void cell_epilogue(const f32x16 (&acc)[4], float (&c)[16], int B, int r0,
                                              int u, int hh, float* __restrict__ gates_t,
                                              float* __restrict__ h_t, float* __restrict__ c_t,
                                              float* h_tile) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = acc_row(r, hh), row = r0 + lr;
        const float ig = sigmoidf_(acc[0][r]);
        const float fg = sigmoidf_(acc[1][r]);
        const float gg = tanhf_(acc[2][r]);
        const float og = sigmoidf_(acc[3][r]);
        c[r] = fg * c[r] + ig * gg;
        const float h = og * tanhf_(c[r]);
        h_tile[lr * PLDA + u] = h;
        if (row < B) {
            const long o = (long)row * PH + u;
            // written once, read again only by the backward pass: keep them out of
            // the L2's way, it holds the weights every workgroup re-reads each step
            __builtin_nontemporal_store(h, h_t + o);
            __builtin_nontemporal_store(c[r], c_t + o);
            if (gates_t) {
                float* gr = gates_t + (long)row * PG4 + u;
                __builtin_nontemporal_store(ig, gr);
                __builtin_nontemporal_store(fg, gr + PH);
                __builtin_nontemporal_store(gg, gr + 2 * PH);
                __builtin_nontemporal_store(og, gr + 3 * PH);
            }
        }
    }
}

// grid = ceil(B/32), block = 256.  P = input dimension of layer 0 (small).
template <int P>
__global__ void __launch_bounds__(256)
lstm2_persistent_forward_kernel(int T, int B, const float* __restrict__ x,      // (T,B,P)
                                const float* __restrict__ Wih0,                  // (4H,P)
                                const float* __restrict__ b0,                    // (4H)
                                const float* __restrict__ Whh0T,                 // (H,4H)
                                const float* __restrict__ Wih1T,                 // (H,4H)
                                const float* __restrict__ Whh1T,                 // (H,4H)
                                const float* __restrict__ b1,                    // (4H)
                                float* __restrict__ gates0, float* __restrict__ h0,
                                float* __restrict__ c0g, float* __restrict__ gates1,
                                float* __restrict__ h1, float* __restrict__ c1g,
                                const float* __restrict__ h_init,   // (2,B,H) or null
                                const float* __restrict__ c_init) { // (2,B,H) or null
    __shared__ float hbuf0[2][32 * PLDA];
    __shared__ float hbuf1[2][32 * PLDA];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
    const int r0 = blockIdx.x * 32, u0 = 32 * wave, u = u0 + j;
    const long BH = (long)B * PH, BG = (long)B * PG4;

    float c0[16], c1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        c0[r] = c1[r] = 0.f;
        const int lr = acc_row(r, hh), row = r0 + lr;
        if (c_init && row < B) {
            c0[r] = c_init[(long)row * PH + u];
            c1[r] = c_init[BH + (long)row * PH + u];
        }
        if (h_init) {   // h of "step -1" goes where step 0 looks for it
            hbuf0[1][lr * PLDA + u] = row < B ? h_init[(long)row * PH + u] : 0.f;
            hbuf1[1][lr * PLDA + u] = row < B ? h_init[BH + (long)row * PH + u] : 0.f;
        }
    }
    if (h_init) __syncthreads();
    const bool carried = h_init != nullptr;
    // this lane's layer-0 input weights and biases (4 gates x P)
    float wx[4][P], bias0[4], bias1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bias0[g] = b0[g * PH + u];
        bias1[g] = b1[g * PH + u];
#pragma unroll
        for (int q = 0; q < P; ++q) wx[g][q] = Wih0[(long)(g * PH + u) * P + q];
    }
    for (int t = 0; t < T; ++t) {
        f32x16 acc[4];
        // ---- layer 0: input projection (P is tiny: VALU) + recurrent product
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + acc_row(r, hh);
            float xv[P];
#pragma unroll
            for (int q = 0; q < P; ++q) xv[q] = row < B ? x[((long)t * B + row) * P + q] : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = bias0[g];
#pragma unroll
                for (int q = 0; q < P; ++q) s += xv[q] * wx[g][q];
                acc[g][r] = s;
            }
        }
        if (t > 0 || carried) mfma_k128_4gates(acc, hbuf0[(t - 1) & 1], Whh0T, u0);
        cell_epilogue(acc, c0, B, r0, u, hh, gates0 ? gates0 + t * BG : nullptr, h0 + t * BH, c0g + t * BH,
                      hbuf0[t & 1]);
        __syncthreads();
        // ---- layer 1: input product with h0_t, recurrent product with h1_{t-1}
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = bias1[g];
        }
        mfma_k128_4gates(acc, hbuf0[t & 1], Wih1T, u0);
        if (t > 0 || carried) mfma_k128_4gates(acc, hbuf1[(t - 1) & 1], Whh1T, u0);
        cell_epilogue(acc, c1, B, r0, u, hh, gates1 ? gates1 + t * BG : nullptr, h1 + t * BH, c1g + t * BH,
                      hbuf1[t & 1]);
        // hbuf1[t&1] is first read after the barrier of step t+1; hbuf0[(t+1)&1] was
        // last read before the barrier above
    }
}

// ---------------------------------------------------------------------------
// 16-row variant of the persistent kernels.  With 32-row tiles a batch of a few
// thousand sequences gives fewer workgroups than the chip has compute units;
// 16-row tiles (v_mfma_f32_16x16x4_f32) double the workgroup count at the price
// of streaming the weights twice as often from L2.  A lane owns rows 4s..4s+3
// (s = lane>>4) and the two adjacent hidden units u0+2c, u0+2c+1 (c = lane&15),
// so weight loads, h/c/gate stores and LDS writes are all 8-byte accesses.
// ---------------------------------------------------------------------------
constexpr int PLDA16 = PH + 4;     // LDS row stride of a 16-row h tile
constexpr int LDG16 = PG4 + 4;     // LDS row stride of a 16-row gate-gradient tile
constexpr int KC16 = 4;            // k-steps (of 4) per prefetch chunk, forward
constexpr int GC16 = 16;           // k-steps (of 4) per prefetch chunk, backward
// (chunks twice as long push the kernels past 256 registers and run 1.6x slower)

// acc[g][hf] += A_tile (16 x 128, LDS) * Bt[:, g*PH + u0 + 2c + hf], Bt = (128, 512).
// The weight operand is streamed from L2 through two register buffers used in
// turn (no copies, so each MFMA waits only for its own load): one chunk covers
// KC16 * 8 MFMAs = 256 * KC16 cycles of L2 latency.
struct WChunk4 {
    float2 v[KC16][4];
    __device__ __forceinline__ void load(const float* __restrict__ wb, int kk0) {
#pragma unroll
        for (int u = 0; u < KC16; ++u) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                v[u][g] = *reinterpret_cast<const float2*>(wb + (long)(4 * (kk0 + u)) * PG4 + g * PH);
        }
    }
    __device__ __forceinline__ void apply(f32x4 (&acc)[4][2], const float* a_row, int kk0) const {
#pragma unroll
        for (int u = 0; u < KC16; ++u) {
            const float a = a_row[4 * (kk0 + u)];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[u][g].x, acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[u][g].y, acc[g][1], 0, 0, 0);
            }
        }
    }
};

// b0 arrives holding the first chunk of Bt (preload16): a product that follows a cell
// epilogue gets its first weights requested BEFORE the epilogue's stores, because loads
// and stores share one in-order counter and a load queued behind ~20 KB of stores per
// wavefront waits for all of them to drain.
__device__ __forceinline__ const float* weight_base16(const float* __restrict__ Bt, int u0) {
    const int lane = threadIdx.x & 63, c = lane & 15, s = lane >> 4;
    return Bt + u0 + 2 * c + (long)s * PG4;
}
__device__ __forceinline__ void mfma16_k128_4gates(f32x4 (&acc)[4][2], const float* a_tile,
                                                   const float* __restrict__ Bt, int u0,
                                                   WChunk4& b0) {
    static_assert((PH / 4) % (2 * KC16) == 0, "chunk pairs must tile K");
    const int lane = threadIdx.x & 63, c = lane & 15, s = lane >> 4;
    const float* wb = weight_base16(Bt, u0);
    const float* a_row = a_tile + c * PLDA16 + s;
    WChunk4 b1;
#pragma unroll 1
    for (int kk0 = 0; kk0 < PH / 4; kk0 += 2 * KC16) {
        b1.load(wb, kk0 + KC16);
        b0.apply(acc, a_row, kk0);
        if (kk0 + 2 * KC16 < PH / 4) b0.load(wb, kk0 + 2 * KC16);
        b1.apply(acc, a_row, kk0 + KC16);
    }
}
__device__ __forceinline__ void mfma16_k128_4gates(f32x4 (&acc)[4][2], const float* a_tile,
                                                   const float* __restrict__ Bt, int u0) {
    WChunk4 b0;
    b0.load(weight_base16(Bt, u0), 0);
    mfma16_k128_4gates(acc, a_tile, Bt, u0, b0);
}

__device__ __forceinline__ void cell_epilogue16(const f32x4 (&acc)[4][2], float (&cst)[4][2], int B,
                                                int r0, int ub, int s,
                                                float* __restrict__ gates_t, float* __restrict__ h_t,
                                                float* __restrict__ c_t, float* h_tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lr = 4 * s + i, row = r0 + lr;
        float2 ig, fg, gg, og, hv, cv;
        ig.x = sigmoidf_(acc[0][0][i]); ig.y = sigmoidf_(acc[0][1][i]);
        fg.x = sigmoidf_(acc[1][0][i]); fg.y = sigmoidf_(acc[1][1][i]);
        gg.x = tanhf_(acc[2][0][i]);     gg.y = tanhf_(acc[2][1][i]);
        og.x = sigmoidf_(acc[3][0][i]); og.y = sigmoidf_(acc[3][1][i]);
        cv.x = cst[i][0] = fg.x * cst[i][0] + ig.x * gg.x;
        cv.y = cst[i][1] = fg.y * cst[i][1] + ig.y * gg.y;
        hv.x = og.x * tanhf_(cv.x);
        hv.y = og.y * tanhf_(cv.y);
        *reinterpret_cast<float2*>(h_tile + lr * PLDA16 + ub) = hv;
        if (row < B) {
            const long o = (long)row * PH + ub;
            store_streaming(h_t + o, hv);
            store_streaming(c_t + o, cv);
            if (gates_t) {
                float* gr = gates_t + (long)row * PG4 + ub;
                store_streaming(gr, ig);
                store_streaming(gr + PH, fg);
                store_streaming(gr + 2 * PH, gg);
                store_streaming(gr + 3 * PH, og);
            }
        }
    }
}

// grid = ceil(B/16), block = 256.
template <int P>
__global__ void __launch_bounds__(256)
lstm2_persistent_forward16_kernel(int T, int B, const float* __restrict__ x,
                                  const float* __restrict__ Wih0, const float* __restrict__ b0,
                                  const float* __restrict__ Whh0T, const float* __restrict__ Wih1T,
                                  const float* __restrict__ Whh1T, const float* __restrict__ b1,
                                  float* __restrict__ gates0, float* __restrict__ h0,
                                  float* __restrict__ c0g, float* __restrict__ gates1,
                                  float* __restrict__ h1, float* __restrict__ c1g,
                                  const float* __restrict__ h_init,   // (2,B,H) or null
                                  const float* __restrict__ c_init) { // (2,B,H) or null
    __shared__ __attribute__((aligned(16))) float hbuf0[2][16 * PLDA16];
    __shared__ __attribute__((aligned(16))) float hbuf1[2][16 * PLDA16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, s = lane >> 4;
    const int r0 = blockIdx.x * 16, u0 = 32 * wave, ub = u0 + 2 * c;
    const long BH = (long)B * PH, BG = (long)B * PG4;

    float c0[4][2], c1[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c0[i][0] = c0[i][1] = c1[i][0] = c1[i][1] = 0.f;
        const int lr = 4 * s + i, row = r0 + lr;
        if (c_init && row < B) {
            const float2 a = *reinterpret_cast<const float2*>(c_init + (long)row * PH + ub);
            const float2 b = *reinterpret_cast<const float2*>(c_init + BH + (long)row * PH + ub);
            c0[i][0] = a.x; c0[i][1] = a.y;
            c1[i][0] = b.x; c1[i][1] = b.y;
        }
        if (h_init) {   // h of "step -1" goes where step 0 looks for it
            float2 a = {0.f, 0.f}, b = {0.f, 0.f};
            if (row < B) {
                a = *reinterpret_cast<const float2*>(h_init + (long)row * PH + ub);
                b = *reinterpret_cast<const float2*>(h_init + BH + (long)row * PH + ub);
            }
            *reinterpret_cast<float2*>(&hbuf0[1][lr * PLDA16 + ub]) = a;
            *reinterpret_cast<float2*>(&hbuf1[1][lr * PLDA16 + ub]) = b;
        }
    }
    if (h_init) __syncthreads();
    const bool carried = h_init != nullptr;
    float wx[4][2][P], bias0[4][2], bias1[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            bias0[g][hf] = b0[g * PH + ub + hf];
            bias1[g][hf] = b1[g * PH + ub + hf];
#pragma unroll
            for (int q = 0; q < P; ++q) wx[g][hf][q] = Wih0[(long)(g * PH + ub + hf) * P + q];
        }
    }
    WChunk4 pre;
    pre.load(weight_base16(Whh0T, u0), 0);
    for (int t = 0; t < T; ++t) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * s + i;
            float xv[P];
#pragma unroll
            for (int q = 0; q < P; ++q) xv[q] = row < B ? x[((long)t * B + row) * P + q] : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    float v = bias0[g][hf];
#pragma unroll
                    for (int q = 0; q < P; ++q) v += xv[q] * wx[g][hf][q];
                    acc[g][hf][i] = v;
                }
            }
        }
        if (t > 0 || carried) mfma16_k128_4gates(acc, hbuf0[(t - 1) & 1], Whh0T, u0, pre);
        pre.load(weight_base16(Wih1T, u0), 0);      // ahead of the epilogue's stores
        cell_epilogue16(acc, c0, B, r0, ub, s, gates0 ? gates0 + t * BG : nullptr, h0 + t * BH, c0g + t * BH,
                        hbuf0[t & 1]);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[g][hf][i] = bias1[g][hf];
            }
        }
        mfma16_k128_4gates(acc, hbuf0[t & 1], Wih1T, u0, pre);
        if (t > 0 || carried) mfma16_k128_4gates(acc, hbuf1[(t - 1) & 1], Whh1T, u0);
        pre.load(weight_base16(Whh0T, u0), 0);      // for the next step, ahead of the stores
        cell_epilogue16(acc, c1, B, r0, ub, s, gates1 ? gates1 + t * BG : nullptr, h1 + t * BH, c1g + t * BH,
                        hbuf1[t & 1]);
        // the barrier of step t+1 orders these h1 writes before their first read,
        // and the barrier above ordered the last reads of hbuf0[(t+1)&1] before now
    }
}

// acc[hf] (16 rows x units u0+2c+hf) += g_tile (16 x 512, LDS) * W[:, unit], W = (512, 128)
struct WChunk1 {
    float2 v[GC16];
    __device__ __forceinline__ void load(const float* __restrict__ wb, int kk0) {
#pragma unroll
        for (int u = 0; u < GC16; ++u)
            v[u] = *reinterpret_cast<const float2*>(wb + (long)(4 * (kk0 + u)) * PH);
    }
    __device__ __forceinline__ void apply(f32x4 (&acc)[2], const float* a_row, int kk0) const {
#pragma unroll
        for (int u = 0; u < GC16; ++u) {
            const float a = a_row[4 * (kk0 + u)];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[u].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[u].y, acc[1], 0, 0, 0);
        }
    }
};

__device__ __forceinline__ void mfma16_k512(f32x4 (&acc)[2], const float* g_tile,
                                            const float* __restrict__ W, int u0) {
    static_assert((PG4 / 4) % (2 * GC16) == 0, "chunk pairs must tile K");
    const int lane = threadIdx.x & 63, c = lane & 15, s = lane >> 4;
    const float* wb = W + u0 + 2 * c + (long)s * PH;
    const float* a_row = g_tile + c * LDG16 + s;
    WChunk1 b0, b1;
    b0.load(wb, 0);
#pragma unroll 1
    for (int kk0 = 0; kk0 < PG4 / 4; kk0 += 2 * GC16) {
        b1.load(wb, kk0 + GC16);
        b0.apply(acc, a_row, kk0);
        if (kk0 + 2 * GC16 < PG4 / 4) b0.load(wb, kk0 + 2 * GC16);
        b1.apply(acc, a_row, kk0 + GC16);
    }
}

// Two products that share the A operand: accA += g_tile * WA[:, unit], accB += g_tile * WB[:, unit]
// (one LDS read feeds four MFMAs on four independent accumulators).
constexpr int GC16D = GC16 / 2;
struct WChunk2 {
    float2 va[GC16D], vb[GC16D];
    __device__ __forceinline__ void load(const float* __restrict__ wa, const float* __restrict__ wb,
                                         int kk0) {
#pragma unroll
        for (int u = 0; u < GC16D; ++u) {
            va[u] = *reinterpret_cast<const float2*>(wa + (long)(4 * (kk0 + u)) * PH);
            vb[u] = *reinterpret_cast<const float2*>(wb + (long)(4 * (kk0 + u)) * PH);
        }
    }
    __device__ __forceinline__ void apply(f32x4 (&accA)[2], f32x4 (&accB)[2], const float* a_row,
                                          int kk0) const {
#pragma unroll
        for (int u = 0; u < GC16D; ++u) {
            const float a = a_row[4 * (kk0 + u)];
            accA[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, va[u].x, accA[0], 0, 0, 0);
            accA[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, va[u].y, accA[1], 0, 0, 0);
            accB[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vb[u].x, accB[0], 0, 0, 0);
            accB[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vb[u].y, accB[1], 0, 0, 0);
        }
    }
};

__device__ __forceinline__ void mfma16_k512_dual(f32x4 (&accA)[2], f32x4 (&accB)[2],
                                                 const float* g_tile, const float* __restrict__ WA,
                                                 const float* __restrict__ WB, int u0) {
    static_assert((PG4 / 4) % (2 * GC16D) == 0, "chunk pairs must tile K");
    const int lane = threadIdx.x & 63, c = lane & 15, s = lane >> 4;
    const long off = u0 + 2 * c + (long)s * PH;
    const float *wa = WA + off, *wb = WB + off;
    const float* a_row = g_tile + c * LDG16 + s;
    WChunk2 b0, b1;
    b0.load(wa, wb, 0);
#pragma unroll 1
    for (int kk0 = 0; kk0 < PG4 / 4; kk0 += 2 * GC16D) {
        b1.load(wa, wb, kk0 + GC16D);
        b0.apply(accA, accB, a_row, kk0);
        if (kk0 + 2 * GC16D < PG4 / 4) b0.load(wa, wb, kk0 + 2 * GC16D);
        b1.apply(accA, accB, a_row, kk0 + GC16D);
    }
}

// What the cell adjoint of one layer and step reads from global memory for this
// lane's 4 rows x 2 units.  None of it depends on the recurrence, so it is loaded
// one phase ahead, while the previous phase's MFMAs run.
struct CellIn {
    float2 ig[4], fg[4], gg[4], og[4], cc[4], cp[4];
    __device__ __forceinline__ void load(int t, int B, int r0, int ub, int s,
                                         const float* __restrict__ gates,
                                         const float* __restrict__ cg) {
        const long BH = (long)B * PH, BG = (long)B * PG4;
        const float2 z = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * s + i;
            ig[i] = fg[i] = gg[i] = og[i] = cc[i] = cp[i] = z;
            if (row < B) {
                const float* gr = gates + t * BG + (long)row * PG4 + ub;
                ig[i] = load_streaming(gr);
                fg[i] = load_streaming(gr + PH);
                gg[i] = load_streaming(gr + 2 * PH);
                og[i] = load_streaming(gr + 3 * PH);
                const long o = (long)row * PH + ub;
                cc[i] = load_streaming(cg + t * BH + o);
                if (t > 0) cp[i] = load_streaming(cg + (t - 1) * BH + o);
            }
        }
    }
};

// element-wise LSTM cell adjoint for this lane's 4 rows x 2 units; dh is the
// total gradient w.r.t. h_t.  Writes dG (global, step t) and the LDS A-operand
// tile, and returns the gate gradients in dgt[row][gate] (zero for rows >= B).
__device__ __forceinline__ void cell_backward16(const float (&dh)[4][2], float (&dc)[4][2],
                                                const CellIn& in, int t, int B, int r0, int ub,
                                                int s, float* __restrict__ dG, float* g_tile,
                                                float2 (&dgt)[4][4]) {
    const long BG = (long)B * PG4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lr = 4 * s + i, row = r0 + lr;
        float2 di, df, dg, dout;
        {
            const float tc = tanhf_(in.cc[i].x);
            const float dct = dh[i][0] * in.og[i].x * (1.f - tc * tc) + dc[i][0];
            di.x = dct * in.gg[i].x * in.ig[i].x * (1.f - in.ig[i].x);
            df.x = dct * in.cp[i].x * in.fg[i].x * (1.f - in.fg[i].x);
            dg.x = dct * in.ig[i].x * (1.f - in.gg[i].x * in.gg[i].x);
            dout.x = dh[i][0] * tc * in.og[i].x * (1.f - in.og[i].x);
            dc[i][0] = dct * in.fg[i].x;
        }
        {
            const float tc = tanhf_(in.cc[i].y);
            const float dct = dh[i][1] * in.og[i].y * (1.f - tc * tc) + dc[i][1];
            di.y = dct * in.gg[i].y * in.ig[i].y * (1.f - in.ig[i].y);
            df.y = dct * in.cp[i].y * in.fg[i].y * (1.f - in.fg[i].y);
            dg.y = dct * in.ig[i].y * (1.f - in.gg[i].y * in.gg[i].y);
            dout.y = dh[i][1] * tc * in.og[i].y * (1.f - in.og[i].y);
            dc[i][1] = dct * in.fg[i].y;
        }
        if (row < B) {
            float* d = dG + t * BG + (long)row * PG4 + ub;
            store_streaming(d, di);
            store_streaming(d + PH, df);
            store_streaming(d + 2 * PH, dg);
            store_streaming(d + 3 * PH, dout);
        }
        float* tl = g_tile + lr * LDG16 + ub;
        *reinterpret_cast<float2*>(tl) = di;
        *reinterpret_cast<float2*>(tl + PH) = df;
        *reinterpret_cast<float2*>(tl + 2 * PH) = dg;
        *reinterpret_cast<float2*>(tl + 3 * PH) = dout;
        dgt[i][0] = di;
        dgt[i][1] = df;
        dgt[i][2] = dg;
        dgt[i][3] = dout;
    }
}

// grid = ceil(B/16), block = 256.  Besides dG0/dG1 (the operands of the
// time-batched recurrent weight-gradient GEMMs) the kernel forms everything whose
// library GEMM would be badly shaped: dx = dG0 * Wih0 (T,B,P), and per-workgroup
// partial sums over (t, rows) of dG0, dG1 (bias gradients) and dG0^T x (layer-0
// input weights), written to partial[blockIdx.x][ 4H | 4H | 4H*P ].
//
// Per step: (A) cell adjoint of layer 1 -> tile1; barrier; (B) tile1 x [Whh1 | Wih1]
// in one pass; (C) cell adjoint of layer 0 -> tile0; barrier; (D) tile0 x Whh0.
// The two tiles are separate so that two barriers per step order every LDS reuse:
// tile1 is rewritten in (A) of the next step, after the barrier of (C) that every
// wavefront reaches only after its reads in (B); tile0 is rewritten in (C) of the
// next step, after the barrier of (A) that follows every wavefront's reads in (D).
template <int P>
__global__ void __launch_bounds__(256)
lstm2_persistent_backward16_kernel(int T, int B, const float* __restrict__ g_h1,
                                   const float* __restrict__ x,
                                   const float* __restrict__ gates0, const float* __restrict__ c0g,
                                   const float* __restrict__ gates1, const float* __restrict__ c1g,
                                   const float* __restrict__ Wih0,
                                   const float* __restrict__ Whh0, const float* __restrict__ Wih1,
                                   const float* __restrict__ Whh1, float* __restrict__ dG0,
                                   float* __restrict__ dG1, float* __restrict__ dx,
                                   float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float tile1[16 * LDG16];
    __shared__ __attribute__((aligned(16))) float tile0[16 * LDG16];
    __shared__ float dx_part[4][16][P];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, s = lane >> 4;
    const int r0 = blockIdx.x * 16, u0 = 32 * wave, ub = u0 + 2 * c;
    const long BH = (long)B * PH;
    float dc0[4][2], dc1[4][2];
    f32x4 dh0rec[2], dh1rec[2];
    float sb0[4][2], sb1[4][2], sw[4][2][P], wx[4][2][P];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dc0[i][0] = dc0[i][1] = dc1[i][0] = dc1[i][1] = 0.f;
        dh0rec[0][i] = dh0rec[1][i] = dh1rec[0][i] = dh1rec[1][i] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            sb0[g][hf] = sb1[g][hf] = 0.f;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                sw[g][hf][q] = 0.f;
                wx[g][hf][q] = Wih0[(long)(g * PH + ub + hf) * P + q];
            }
        }
    }
    auto load_gh = [&](int t, float2 (&gh)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * s + i;
            gh[i] = make_float2(0.f, 0.f);
            if (row < B) gh[i] = *reinterpret_cast<const float2*>(g_h1 + t * BH + (long)row * PH + ub);
        }
    };
    CellIn in1, in0;
    float2 gh[4];
    float xv[4][P];
    in1.load(T - 1, B, r0, ub, s, gates1, c1g);
    load_gh(T - 1, gh);
    for (int t = T - 1; t >= 0; --t) {
        float dh[4][2];
        float2 dgt[4][4];
        // ---- (A) layer 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dh[i][0] = dh1rec[0][i] + gh[i].x;
            dh[i][1] = dh1rec[1][i] + gh[i].y;
        }
        cell_backward16(dh, dc1, in1, t, B, r0, ub, s, dG1, tile1, dgt);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sb1[g][0] += dgt[i][g].x;
                sb1[g][1] += dgt[i][g].y;
            }
        }
        // inputs of (C), in flight during (B)
        in0.load(t, B, r0, ub, s, gates0, c0g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 4 * s + i;
#pragma unroll
            for (int q = 0; q < P; ++q) xv[i][q] = row < B ? x[((long)t * B + row) * P + q] : 0.f;
        }
        __syncthreads();
        // ---- (B) dh1rec = tile1 x Whh1 (recurrent path of layer 1, for step t-1);
        //          dh0rec += tile1 x Wih1 (layer 1's input path into layer 0's h_t)
#pragma unroll
        for (int i = 0; i < 4; ++i) dh1rec[0][i] = dh1rec[1][i] = 0.f;
        mfma16_k512_dual(dh1rec, dh0rec, tile1, Whh1, Wih1, u0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dh[i][0] = dh0rec[0][i];
            dh[i][1] = dh0rec[1][i];
        }
        // ---- (C) layer 0
        cell_backward16(dh, dc0, in0, t, B, r0, ub, s, dG0, tile0, dgt);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dxp[P];
#pragma unroll
            for (int q = 0; q < P; ++q) dxp[q] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                sb0[g][0] += dgt[i][g].x;
                sb0[g][1] += dgt[i][g].y;
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    sw[g][0][q] += dgt[i][g].x * xv[i][q];
                    sw[g][1][q] += dgt[i][g].y * xv[i][q];
                    dxp[q] += dgt[i][g].x * wx[g][0][q] + dgt[i][g].y * wx[g][1][q];
                }
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const float v = row_sum16(dxp[q]);
                if (c == 0) dx_part[wave][4 * s + i][q] = v;
            }
        }
        // inputs of the next step's (A), in flight during (D)
        if (t > 0) {
            in1.load(t - 1, B, r0, ub, s, gates1, c1g);
            load_gh(t - 1, gh);
        }
        __syncthreads();
        if (threadIdx.x < 16 * P) {
            const int lr = threadIdx.x / P, q = threadIdx.x % P;
            if (r0 + lr < B)
                dx[((long)t * B + r0 + lr) * P + q] = dx_part[0][lr][q] + dx_part[1][lr][q] +
                                                      dx_part[2][lr][q] + dx_part[3][lr][q];
        }
        // ---- (D) dh0rec = tile0 x Whh0 (recurrent path of layer 0, for step t-1)
#pragma unroll
        for (int i = 0; i < 4; ++i) dh0rec[0][i] = dh0rec[1][i] = 0.f;
        if (t > 0) mfma16_k512(dh0rec, tile0, Whh0, u0);
    }
    // sum the per-lane column accumulators over the four row groups (lane bits 4, 5)
    // both bias sums are written twice: torch.nn.LSTM has two bias vectors per layer with
    // the same gradient, and each then gets memory of its own without a copy
    float* part = partial + (long)blockIdx.x * (PG4 * (4 + P));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int col = g * PH + ub + hf;
            float a = sb0[g][hf], b = sb1[g][hf];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (s == 0) {
                part[col] = a;
                part[PG4 + col] = a;
                part[2 * PG4 + col] = b;
                part[3 * PG4 + col] = b;
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
                float w = sw[g][hf][q];
                w += __shfl_xor(w, 16, 64);
                w += __shfl_xor(w, 32, 64);
                if (s == 0) part[4 * PG4 + (long)col * P + q] = w;
            }
        }
    }
}

// Forward: 16-row tiles while 32-row tiles would leave some of the 256 compute
// units without a workgroup (32-row tiles stream the weights half as often and win
// once the chip is full).  KVAE_LSTM_TILE_ROWS=16|32 forces one form (tests, timing).
// The backward kernel exists in the 16-row form only: it is the faster one at every
// batch size measured.
inline bool lstm2_use_16_row_tiles(int B) {
    const char* e = getenv("KVAE_LSTM_TILE_ROWS");
    const int forced = e ? atoi(e) : 0;
    if (forced == 16) return true;
    if (forced == 32) return false;
    // 32-row tiles stream the weights half as often but halve the workgroup count; they
    // win only where the 16-row grid would need a second pass over the 256 CUs and
    // the 32-row grid does not (measured: 4096 < B <= 8192)
    const int tiles16 = (B + 15) / 16;
    return !(tiles16 > 256 && tiles16 <= 512);
}

// Operands of the persistent forward kernel from the module's parameters in one
// launch: the three (4H,H) matrices transposed to (H,4H) and the two summed bias
// vectors, packed as [Whh0T | Wih1T | Whh1T | b0 | b1].  Five library launches
// (three transposed copies, two additions) otherwise, every training step.
constexpr int PREP_TILES = (PG4 / 32) * (PH / 32);
__global__ void __launch_bounds__(256)
lstm2_prepare_kernel(const float* __restrict__ w_hh0, const float* __restrict__ w_ih1,
                     const float* __restrict__ w_hh1, const float* __restrict__ b_ih0,
                     const float* __restrict__ b_hh0, const float* __restrict__ b_ih1,
                     const float* __restrict__ b_hh1, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (b == 3 * PREP_TILES) {
        float* b0 = out + 3 * PH * PG4;
        for (int e = threadIdx.x; e < PG4; e += 256) {
            b0[e] = b_ih0[e] + b_hh0[e];
            b0[PG4 + e] = b_ih1[e] + b_hh1[e];
        }
        return;
    }
    const int m = b / PREP_TILES, t = b % PREP_TILES, tr = t / (PH / 32), tc = t % (PH / 32);
    const float* src = m == 0 ? w_hh0 : (m == 1 ? w_ih1 : w_hh1);
    float* dst = out + (long)m * PH * PG4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        tile[r][tx] = src[(32 * tr + r) * PH + 32 * tc + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 8 * i;
        dst[(32 * tc + r) * PG4 + 32 * tr + tx] = tile[tx][r];
    }
}

// One time step of one layer's adjoint, element by element (no matrix product): the gate
// pre-activation gradients and the gradient handed to the previous step's cell state from
// the gradient w.r.t. this step's h (the sum of up to two tensors: what the layer above or
// the head sends, and the recurrent part dG_{t+1} W_hh) and the cell-state gradient arriving
// from step t+1.  gates (B,4H) activated i,f,g,o as the forward kernels store them.
// grid = ceil(B * H / 4 / 256), block = 256; four adjacent units per lane.
template <int H>
__global__ void __launch_bounds__(256)
lstm128_cell_step_backward_kernel(long B, const float* __restrict__ gates, const float* __restrict__ c_prev,
                          const float* __restrict__ c, const float* __restrict__ dh_a,
                          const float* __restrict__ dh_b, const float* __restrict__ dc_in,
                          float* __restrict__ dG, float* __restrict__ dc_prev,
                          const float* __restrict__ Wih, int P, const float* __restrict__ dx_add,
                          float* __restrict__ dx) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= B * (H / 4)) return;          // whole rows: the H / 4 = 32 lanes of a row leave together
    const long b = q / (H / 4);
    const int u = 4 * (int)(q % (H / 4));
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](const float* ptr, long at) {
        return ptr ? *reinterpret_cast<const float4*>(ptr + at) : zero;
    };
    const float4 gi = ld(gates, b * 4 * H + u), gf = ld(gates, b * 4 * H + H + u),
                 gg = ld(gates, b * 4 * H + 2 * H + u), go = ld(gates, b * 4 * H + 3 * H + u);
    const float4 cp = ld(c_prev, b * H + u), cc = ld(c, b * H + u);
    const float4 da = ld(dh_a, b * H + u), db = ld(dh_b, b * H + u), dcn = ld(dc_in, b * H + u);
    float4 oi, of, og, oo, ocp;
    auto one = [&](float i, float f, float g, float o, float cprev, float cnow, float dh, float dc,
                   float& di, float& df, float& dg, float& d_o, float& dcp) {
        const float tc = tanhf_(cnow);
        const float dct = dh * o * (1.f - tc * tc) + dc;
        di = dct * g * i * (1.f - i);
        df = dct * cprev * f * (1.f - f);
        dg = dct * i * (1.f - g * g);
        d_o = dh * tc * o * (1.f - o);
        dcp = dct * f;
    };
    one(gi.x, gf.x, gg.x, go.x, cp.x, cc.x, da.x + db.x, dcn.x, oi.x, of.x, og.x, oo.x, ocp.x);
    one(gi.y, gf.y, gg.y, go.y, cp.y, cc.y, da.y + db.y, dcn.y, oi.y, of.y, og.y, oo.y, ocp.y);
    one(gi.z, gf.z, gg.z, go.z, cp.z, cc.z, da.z + db.z, dcn.z, oi.z, of.z, og.z, oo.z, ocp.z);
    one(gi.w, gf.w, gg.w, go.w, cp.w, cc.w, da.w + db.w, dcn.w, oi.w, of.w, og.w, oo.w, ocp.w);
    *reinterpret_cast<float4*>(dG + b * 4 * H + u) = oi;
    *reinterpret_cast<float4*>(dG + b * 4 * H + H + u) = of;
    *reinterpret_cast<float4*>(dG + b * 4 * H + 2 * H + u) = og;
    *reinterpret_cast<float4*>(dG + b * 4 * H + 3 * H + u) = oo;
    *reinterpret_cast<float4*>(dc_prev + b * H + u) = ocp;
    if (dx) {
        // gradient w.r.t. a narrow layer input: dx[b] = dG[b] Wih (+ dx_add[b]), Wih (4H,P);
        // the 32 lanes of a row (half a wavefront) add their parts
        static_assert(H == 128, "a row is 32 lanes");
        const float4 parts[4] = {oi, of, og, oo};
        for (int qq = 0; qq < P; ++qq) {
            float acc = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* w = Wih + (long)(g * H + u) * P + qq;
                acc += parts[g].x * w[0] + parts[g].y * w[P] + parts[g].z * w[2 * P] + parts[g].w * w[3 * P];
            }
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 32);
            if ((threadIdx.x & 31) == 0) dx[b * P + qq] = acc + (dx_add ? dx_add[b * P + qq] : 0.f);
        }
    }
}

}  // namespace kvae

extern "C" {

long kvae_lstm2_prepared_elems(void) { return 3L * kvae::PH * kvae::PG4 + 2L * kvae::PG4; }

int kvae_lstm2_prepare(const void* w_hh0, const void* w_ih1, const void* w_hh1, const void* b_ih0,
                       const void* b_hh0, const void* b_ih1, const void* b_hh1, void* out,
                       void* stream) {
    if (!w_hh0 || !w_ih1 || !w_hh1 || !b_ih0 || !b_hh0 || !b_ih1 || !b_hh1 || !out)
        return KVAE_E_BADARG;
    hipLaunchKernelGGL(kvae::lstm2_prepare_kernel, dim3(3 * kvae::PREP_TILES + 1), dim3(256), 0,
                       (hipStream_t)stream, (const float*)w_hh0, (const float*)w_ih1,
                       (const float*)w_hh1, (const float*)b_ih0, (const float*)b_hh0,
                       (const float*)b_ih1, (const float*)b_hh1, (float*)out);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}


int kvae_lstm2_forward(int T, int B, int P, const void* x, const void* Wih0, const void* b0,
                       const void* Whh0T, const void* Wih1T, const void* Whh1T, const void* b1,
                       void* gates0, void* h0, void* c0, void* gates1, void* h1, void* c1,
                       const void* h_init, const void* c_init, void* stream) {
    if (T < 1 || B < 1 || !x || !Wih0 || !b0 || !Whh0T || !Wih1T || !Whh1T || !b1 || !h0 || !c0 ||
        !h1 || !c1 || (gates0 == nullptr) != (gates1 == nullptr) ||
        (h_init == nullptr) != (c_init == nullptr))
        return KVAE_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool rows16 = kvae::lstm2_use_16_row_tiles(B);
    const dim3 grid(rows16 ? (B + 15) / 16 : (B + 31) / 32);
#define KVAE_L2F_ARGS                                                                           \
    T, B, (const float*)x, (const float*)Wih0, (const float*)b0, (const float*)Whh0T,           \
        (const float*)Wih1T, (const float*)Whh1T, (const float*)b1, (float*)gates0, (float*)h0, \
        (float*)c0, (float*)gates1, (float*)h1, (float*)c1, (const float*)h_init,               \
        (const float*)c_init
#define KVAE_L2F(P_)                                                                            \
    if (rows16)                                                                                 \
        hipLaunchKernelGGL(kvae::lstm2_persistent_forward16_kernel<P_>, grid, dim3(256), 0, s,  \
                           KVAE_L2F_ARGS);                                                      \
    else                                                                                        \
        hipLaunchKernelGGL(kvae::lstm2_persistent_forward_kernel<P_>, grid, dim3(256), 0, s,    \
                           KVAE_L2F_ARGS)
    switch (P) {
        case 1: KVAE_L2F(1); break;
        case 2: KVAE_L2F(2); break;
        case 3: KVAE_L2F(3); break;
        case 4: KVAE_L2F(4); break;
        default: return KVAE_E_UNSUPPORTED;
    }
#undef KVAE_L2F
#undef KVAE_L2F_ARGS
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_lstm2_backward_blocks(int B) { return B < 1 ? 0 : (B + 15) / 16; }

int kvae_lstm2_backward(int T, int B, int P, const void* g_h1, const void* x, const void* gates0,
                        const void* c0, const void* gates1, const void* c1, const void* Wih0,
                        const void* Whh0, const void* Wih1, const void* Whh1, void* dG0, void* dG1,
                        void* dx, void* partial, void* stream) {
    if (T < 1 || B < 1 || !g_h1 || !x || !gates0 || !c0 || !gates1 || !c1 || !Wih0 || !Whh0 ||
        !Wih1 || !Whh1 || !dG0 || !dG1 || !dx || !partial)
        return KVAE_E_BADARG;
#define KVAE_L2B(P_)                                                                             \
    hipLaunchKernelGGL(kvae::lstm2_persistent_backward16_kernel<P_>, dim3((B + 15) / 16),        \
                       dim3(256), 0, (hipStream_t)stream, T, B, (const float*)g_h1,              \
                       (const float*)x, (const float*)gates0, (const float*)c0,                  \
                       (const float*)gates1, (const float*)c1, (const float*)Wih0,               \
                       (const float*)Whh0, (const float*)Wih1, (const float*)Whh1, (float*)dG0,  \
                       (float*)dG1, (float*)dx, (float*)partial)
    switch (P) {
        case 1: KVAE_L2B(1); break;
        case 2: KVAE_L2B(2); break;
        case 3: KVAE_L2B(3); break;
        case 4: KVAE_L2B(4); break;
        default: return KVAE_E_UNSUPPORTED;
    }
#undef KVAE_L2B
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_lstm128_cell_backward(long B, const void* gates, const void* c_prev, const void* c,
                               const void* dh_a, const void* dh_b, const void* dc_in, void* dG,
                               void* dc_prev, const void* Wih, int P, const void* dx_add, void* dx,
                               void* stream) {
    if (B < 1 || !gates || !c || !dh_a || !dG || !dc_prev) return KVAE_E_BADARG;
    if (dx && (!Wih || P < 1 || P > 8)) return KVAE_E_BADARG;
    const uintptr_t all = (uintptr_t)gates | (uintptr_t)c_prev | (uintptr_t)c | (uintptr_t)dh_a |
                          (uintptr_t)dh_b | (uintptr_t)dc_in | (uintptr_t)dG | (uintptr_t)dc_prev;
    if (all & 15) return KVAE_E_BADARG;
    const long quads = B * (128 / 4);
    hipLaunchKernelGGL((kvae::lstm128_cell_step_backward_kernel<128>), dim3((unsigned)((quads + 255) / 256)),
                       dim3(256), 0, (hipStream_t)stream, B, (const float*)gates,
                       (const float*)c_prev, (const float*)c, (const float*)dh_a,
                       (const float*)dh_b, (const float*)dc_in, (float*)dG, (float*)dc_prev,
                       (const float*)Wih, P, (const float*)dx_add, (float*)dx);
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
