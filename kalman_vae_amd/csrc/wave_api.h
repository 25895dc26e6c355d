// Entry points of the wavefront-per-sequence translation unit
// (wave_kernels.hip), called by the C ABI when no lane-per-sequence kernel is
// compiled for the requested (n, p, K).  Same contracts as the kvae_* functions
// of include/kvae_kalman.h; KVAE_E_UNSUPPORTED when (n, p, K, dtype) is not in
// the wave list either.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/kvae_kalman.h"
#include "kalman_core.h"

namespace kvae {
int wave_supported(int n, int p, int K, int dtype);
size_t wave_workspace_bytes(const kvae_dims* d, int op);
int wave_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s);
int wave_filter_backward(const kvae_dims* d, const kvae_filter_bwd* io, void* ws, size_t ws_bytes,
                         hipStream_t s);
int wave_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s);
int wave_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws, size_t ws_bytes,
                         hipStream_t s);
int wave_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, hipStream_t s);
int wave_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws, size_t ws_bytes,
                       hipStream_t s);

// Fused objective (loss_kernels.hip): workspace of kvae_kalman_loss_forward / _backward
// (op = KVAE_OP_LOSS_FWD / KVAE_OP_LOSS_BWD), 0 when (n, p, K) has no such kernel.
size_t loss_workspace_bytes(const kvae_dims* d, int op);

// Row form of the fused objective (loss_row_kernels.hip: one sequence over 16 lanes, per-step
// state in LDS, fp32, n and p up to 4).  row_loss_backward recomputes the forward pass and so
// needs nothing saved; with_value it also writes the forward call's outputs.
long row_loss_rows(long B);
long row_loss_resident_sequences(int T);
int row_loss_supported(int n, int p, int K, int dtype, int T, int grad);
size_t row_loss_workspace_bytes(const kvae_dims* d);
int row_loss_forward(const kvae_dims* d, const kvae_loss_fwd* io, void* ws, size_t ws_bytes,
                     hipStream_t s);
int row_loss_backward(const kvae_dims* d, const kvae_loss_bwd* io, void* ws, size_t ws_bytes,
                      hipStream_t s, int with_value);

// Register-tile kernels (tile_kernels.hip): fp32, n and p multiples of 8 up to 32.
// KVAE_E_UNSUPPORTED means "use the LDS kernels".
int tile_supported(int n, int p, int dtype);
int tile_filter_forward(const kvae_dims* d, const kvae_filter_fwd* io, hipStream_t s);
int tile_filter_backward_main(const kvae_dims* d, const FilterBwdArgs<float>& g, hipStream_t s);
int tile_smooth_forward(const kvae_dims* d, const kvae_smooth_fwd* io, hipStream_t s);
int tile_smooth_backward(const kvae_dims* d, const kvae_smooth_bwd* io, void* ws, size_t ws_bytes,
                         hipStream_t s);
int tile_elbo_forward(const kvae_dims* d, const kvae_elbo_fwd* io, hipStream_t s);
// rows of parameter-gradient sums per sequence that tile_elbo_backward writes into its workspace
int tile_elbo_backward_pieces(const kvae_dims* d);
int tile_elbo_backward(const kvae_dims* d, const kvae_elbo_bwd* io, void* ws, size_t ws_bytes,
                       hipStream_t s);
}  // namespace kvae
