// Helpers shared by the LSTM translation units (lstm_kernels.hip, lstm_resident.hip,
// lstm_step_fused.hip, rowstream_matmul.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "device_info.h"

namespace kvae {

// Sum over each row of 16 lanes, in every lane of the row: four DPP butterfly stages
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), no LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float row_exchange16(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += row_exchange16<0xB1>(v);
    v += row_exchange16<0x4E>(v);
    v += row_exchange16<0x141>(v);
    v += row_exchange16<0x140>(v);
    return v;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));

// Gate activations on the hardware exp2/rcp units (about 1 ulp each).  The tanh
// form has an absolute error near 1e-7 around 0, which is what fp32 gate sums
// carry anyway; both saturate cleanly (exp2 -> inf gives rcp -> 0).
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008f * x));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Streaming accesses for tensors that pass through once (activations saved for the
// backward pass, gate gradients): keeping them out of the L2's way leaves it to the
// weights every workgroup re-reads each step.
__device__ __forceinline__ void store_streaming(float* p, float2 v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x2*>(p));
}
__device__ __forceinline__ float2 load_streaming(const float* p) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 w = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
    return make_float2(w[0], w[1]);
}


// Launchers of the register-resident layer kernels (lstm_resident.hip is compiled four
// times: exact fp32 or split-bf16 products, one tile or many tiles per workgroup; the
// many-tiles builds hold the forward kernel only).
#define KVAE_LSTM128_FWD_ARGS                                                                   \
    const float *x, const float *Wih, const float *bias, const float *WhhT, float *gates,       \
        float *h, float *c, const float *h_init, const float *c_init, int store_gates,          \
        hipStream_t stream
#define KVAE_LSTM128_BWD_ARGS                                                                   \
    const float *x, const float *Wih, const float *gates, const float *c, const float *c_init,  \
        const float *g_h, const float *Whh, float *dG, float *dx, float *partial, int KH,       \
        const float *alpha, const float *g_alpha, const float *Wh, float *head_partial,         \
        hipStream_t stream
int launch_lstm128_forward_tiles(int T, int B, int P, int workgroups, KVAE_LSTM128_FWD_ARGS);
int launch_lstm128_forward_tiles_split(int T, int B, int P, int workgroups, KVAE_LSTM128_FWD_ARGS);
int launch_lstm128_forward_exact(int T, int B, int P, KVAE_LSTM128_FWD_ARGS);
int launch_lstm128_forward_split(int T, int B, int P, KVAE_LSTM128_FWD_ARGS);
int launch_lstm128_backward_exact(int T, int B, int P, KVAE_LSTM128_BWD_ARGS);
int launch_lstm128_backward_split(int T, int B, int P, KVAE_LSTM128_BWD_ARGS);

}  // namespace kvae
