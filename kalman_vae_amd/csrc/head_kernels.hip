// Softmax head of the dynamics-parameter network on gfx950:
//   alpha = softmax(h W^T + b),  h (R,H) with R = T*B rows, W (K,H), K <= 8
// (kvae/dynamics_parameter_network.py:57-58 for the LSTM model, 84-90 for the
// last MLP layer).  As library GEMMs these are badly shaped: the weight gradient
// is a (K x R) x (R x H) product with K = 3 and R = 81920, which rocBLAS runs in
// 223 us; here it is one streaming pass over h.
//
// Mapping: 16 lanes per row, lane c holds the V = H/16 contiguous elements
// c*V .. c*V+V-1 of the row (so a row is one contiguous segment across the 16
// lanes and a wavefront covers 4 rows); the K dot products are reduced over the
// 16 lanes with shuffles.  Forward reads h once; backward reads h, alpha and the
// gradient w.r.t. alpha once, writes d_h, and leaves per-workgroup partial sums
// of dW and db that the caller adds up (no float atomics: results are
// deterministic).
#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"

namespace kvae {

constexpr int HEAD_BLOCK = 256;           // threads; 16 rows per pass
constexpr int HEAD_FWD_MAX_BLOCKS = 2048;
constexpr int HEAD_BWD_MAX_BLOCKS = 512;  // = rows of the partial-sum matrix the caller adds up

template <int V>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&v)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
        for (int i = 0; i < V; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(p + i);
            v[i] = q.x; v[i + 1] = q.y; v[i + 2] = q.z; v[i + 3] = q.w;
        }
    } else {
        const float2 q = *reinterpret_cast<const float2*>(p);
        v[0] = q.x; v[1] = q.y;
    }
}

template <int V>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&v)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
        for (int i = 0; i < V; i += 4)
            *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    } else {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    }
}

// Sum over each row of 16 lanes, in every lane of the row: four DPP butterfly stages
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), no LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float row_exchange(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum16(float v) {
    v += row_exchange<0xB1>(v);
    v += row_exchange<0x4E>(v);
    v += row_exchange<0x141>(v);
    v += row_exchange<0x140>(v);
    return v;
}

template <int H, int K>
__global__ void __launch_bounds__(HEAD_BLOCK)
softmax_head_forward_kernel(long R, const float* __restrict__ h, const float* __restrict__ W,
                            const float* __restrict__ b, float* __restrict__ alpha, long lead_rows) {
    constexpr int V = H / 16;
    const int c = threadIdx.x & 15;
    // rows 0 .. lead_rows-1 of the output are the uniform weights 1/K (the filter's
    // weights before the first observation, kvae/state_space_model.py:225)
    for (long i = (long)blockIdx.x * HEAD_BLOCK + threadIdx.x; i < lead_rows * K;
         i += (long)gridDim.x * HEAD_BLOCK)
        alpha[i] = 1.f / (float)K;
    alpha += lead_rows * K;
    float w[K][V], bias[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bias[k] = b[k];
        load_row<V>(W + (long)k * H + c * V, w[k]);
    }
    // every lane of a wavefront runs the same number of passes (shuffles need all
    // 64 lanes), rows past the end are computed on zeros and not stored
    const long groups = (R + 15) / 16;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long row = g * 16 + (threadIdx.x >> 4);
        float hv[V];
        if (row < R) {
            load_row<V>(h + row * H + c * V, hv);
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) hv[v] = 0.f;
        }
        float s[K], mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float a = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) a += hv[v] * w[k][v];
            s[k] = sum16(a) + bias[k];
            mx = fmaxf(mx, s[k]);
        }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            s[k] = expf(s[k] - mx);
            den += s[k];
        }
        const float inv = 1.f / den;
        float out = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) out = (c == k) ? s[k] * inv : out;
        if (row < R && c < K) alpha[row * K + c] = out;
    }
}

template <int H, int K>
__global__ void __launch_bounds__(HEAD_BLOCK)
softmax_head_backward_kernel(long R, const float* __restrict__ h, const float* __restrict__ W,
                             const float* __restrict__ alpha, const float* __restrict__ g_alpha,
                             float* __restrict__ d_h, float* __restrict__ partial) {
    constexpr int V = H / 16;
    constexpr int NP = K * (H + 1);
    __shared__ float red[4][NP];
    const int c = threadIdx.x & 15, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float w[K][V], sw[K][V], sb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        sb[k] = 0.f;
        load_row<V>(W + (long)k * H + c * V, w[k]);
#pragma unroll
        for (int v = 0; v < V; ++v) sw[k][v] = 0.f;
    }
    const long groups = (R + 15) / 16;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long row = g * 16 + (threadIdx.x >> 4);
        if (row >= R) continue;       // no cross-lane traffic inside the loop
        float hv[V], dl[K], dot = 0.f;
        load_row<V>(h + row * H + c * V, hv);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            dl[k] = alpha[row * K + k];
            dot += dl[k] * g_alpha[row * K + k];
        }
        float dh[V];
#pragma unroll
        for (int v = 0; v < V; ++v) dh[v] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            dl[k] *= g_alpha[row * K + k] - dot;      // softmax adjoint
            sb[k] += dl[k];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                dh[v] += dl[k] * w[k][v];
                sw[k][v] += dl[k] * hv[v];
            }
        }
        store_row<V>(d_h + row * H + c * V, dh);
    }
    // sum over the wavefront's four row slots (lane bits 4, 5), then over wavefronts
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float a = sb[k];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (lane == 0) red[wave][K * H + k] = a;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float x = sw[k][v];
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            if (lane < 16) red[wave][k * H + c * V + v] = x;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NP; i += HEAD_BLOCK)
        partial[(long)blockIdx.x * NP + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

inline int head_blocks(long R, int cap) {
    const long groups = (R + 15) / 16;
    return (int)(groups < cap ? groups : cap);
}

template <int H>
int launch_head_forward(long R, int K, const float* h, const float* W, const float* b, float* alpha,
                        long lead_rows, hipStream_t s) {
    const dim3 grid(head_blocks(R, HEAD_FWD_MAX_BLOCKS)), block(HEAD_BLOCK);
    switch (K) {
#define KVAE_HF(K_)                                                                              \
    case K_:                                                                                     \
        hipLaunchKernelGGL((softmax_head_forward_kernel<H, K_>), grid, block, 0, s, R, h, W, b,  \
                           alpha, lead_rows);                                                    \
        return KVAE_OK;
        KVAE_HF(1) KVAE_HF(2) KVAE_HF(3) KVAE_HF(4) KVAE_HF(5) KVAE_HF(6) KVAE_HF(7) KVAE_HF(8)
#undef KVAE_HF
    }
    return KVAE_E_UNSUPPORTED;
}

template <int H>
int launch_head_backward(long R, int K, const float* h, const float* W, const float* alpha,
                         const float* g, float* d_h, float* partial, hipStream_t s) {
    const dim3 grid(head_blocks(R, HEAD_BWD_MAX_BLOCKS)), block(HEAD_BLOCK);
    switch (K) {
#define KVAE_HB(K_)                                                                              \
    case K_:                                                                                     \
        hipLaunchKernelGGL((softmax_head_backward_kernel<H, K_>), grid, block, 0, s, R, h, W,    \
                           alpha, g, d_h, partial);                                              \
        return KVAE_OK;
        KVAE_HB(1) KVAE_HB(2) KVAE_HB(3) KVAE_HB(4) KVAE_HB(5) KVAE_HB(6) KVAE_HB(7) KVAE_HB(8)
#undef KVAE_HB
    }
    return KVAE_E_UNSUPPORTED;
}

}  // namespace kvae

extern "C" {

int kvae_softmax_head_backward_blocks(long R) {
    return R < 1 ? 0 : kvae::head_blocks(R, kvae::HEAD_BWD_MAX_BLOCKS);
}

int kvae_softmax_head_forward(long R, int H, int K, const void* h, const void* W, const void* b,
                              void* alpha, long lead_rows, void* stream) {
    if (R < 1 || lead_rows < 0 || !h || !W || !b || !alpha) return KVAE_E_BADARG;
    const float *hp = (const float*)h, *wp = (const float*)W, *bp = (const float*)b;
    float* ap = (float*)alpha;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (H) {
        case 32: rc = kvae::launch_head_forward<32>(R, K, hp, wp, bp, ap, lead_rows, s); break;
        case 64: rc = kvae::launch_head_forward<64>(R, K, hp, wp, bp, ap, lead_rows, s); break;
        case 128: rc = kvae::launch_head_forward<128>(R, K, hp, wp, bp, ap, lead_rows, s); break;
        case 256: rc = kvae::launch_head_forward<256>(R, K, hp, wp, bp, ap, lead_rows, s); break;
        default: return KVAE_E_UNSUPPORTED;
    }
    if (rc != KVAE_OK) return rc;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

int kvae_softmax_head_backward(long R, int H, int K, const void* h, const void* W,
                               const void* alpha, const void* g_alpha, void* d_h, void* partial,
                               void* stream) {
    if (R < 1 || !h || !W || !alpha || !g_alpha || !d_h || !partial) return KVAE_E_BADARG;
    const float *hp = (const float*)h, *wp = (const float*)W, *ap = (const float*)alpha,
                *gp = (const float*)g_alpha;
    float *dp = (float*)d_h, *pp = (float*)partial;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (H) {
        case 32: rc = kvae::launch_head_backward<32>(R, K, hp, wp, ap, gp, dp, pp, s); break;
        case 64: rc = kvae::launch_head_backward<64>(R, K, hp, wp, ap, gp, dp, pp, s); break;
        case 128: rc = kvae::launch_head_backward<128>(R, K, hp, wp, ap, gp, dp, pp, s); break;
        case 256: rc = kvae::launch_head_backward<256>(R, K, hp, wp, ap, gp, dp, pp, s); break;
        default: return KVAE_E_UNSUPPORTED;
    }
    if (rc != KVAE_OK) return rc;
    return hipGetLastError() == hipSuccess ? KVAE_OK : KVAE_E_LAUNCH;
}

}  // extern "C"
