"""ctypes mirror of include/kvae_kalman.h and the loader of the HIP library.

The library (kalman_vae_amd/csrc/libkvae_kalman.so, built by csrc/Makefile or
``__graft_entry__.build()``) is the only compute backend.  There is no CPU
path: if the library is missing or a tensor is not on a GPU the call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# KVAE_LIB_PATH: another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("KVAE_LIB_PATH") or os.path.join(_HERE, "csrc", "libkvae_kalman.so")

KVAE_F32, KVAE_F64 = 0, 1
OP_FILTER_BWD, OP_SMOOTH_BWD, OP_ELBO_BWD, OP_LOSS_FWD, OP_LOSS_BWD = 1, 2, 3, 4, 5
OP_LOSS_VALUE_AND_GRAD = 6

STATUS_BITS = {
    1: "innovation covariance C P C^T + R is not positive definite",
    2: "predicted covariance is not positive definite",
    4: "smoothed covariance is not positive definite",
    8: "Q, R or the initial covariance is not positive definite",
}

_vp = C.c_void_p


class Dims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ("T", "B", "n", "p", "K", "dtype", "symmetrize", "burn_in")]


def _ptr_struct(name, fields):
    """Structure whose members are all void* except those given as (name, ctype)."""
    fl = []
    for f in fields:
        fl.append(f if isinstance(f, tuple) else (f, _vp))
    return type(name, (C.Structure,), {"_fields_": fl})


FilterFwd = _ptr_struct("FilterFwd", [
    "a", "alpha", "mask", "A_K", "C_K", "Q", "R", "m_in", "P_in",
    ("carry_per_sequence", C.c_int32), ("_pad", C.c_int32),
    "mean_filt", "cov_filt", "mean_pred", "cov_pred", "mat_As", "mat_Cs", "status"])
FilterBwd = _ptr_struct("FilterBwd", [
    ("fwd", FilterFwd), "g_mean_filt", "g_cov_filt", "g_mean_pred", "g_cov_pred",
    "g_mat_As", "g_mat_Cs", "d_a", "d_alpha", "d_m_in", "d_P_in",
    "d_A_K", "d_C_K", "d_Q", "d_R"])
SmoothFwd = _ptr_struct("SmoothFwd", [
    "mean_filt", "cov_filt", "mean_pred", "cov_pred", "mat_As", "mat_Cs", "LR",
    "eps_state", "eps_obs", "mean_smooth", "cov_smooth", "zs", "as_resampled", "status"])
SmoothBwd = _ptr_struct("SmoothBwd", [
    ("fwd", SmoothFwd), "g_mean_smooth", "g_cov_smooth", "g_zs", "g_as_resampled",
    "d_mean_filt", "d_cov_filt", "d_mean_pred", "d_cov_pred", "d_mat_As", "d_mat_Cs", "d_LR"])
ElboFwd = _ptr_struct("ElboFwd", [
    "a", "mean_smooth", "cov_smooth", "mat_As", "mat_Cs", "Q", "R", "m0", "P0", "eps",
    "obs", "trans", "post", "z", "status"])
ElboBwd = _ptr_struct("ElboBwd", [
    ("fwd", ElboFwd), "g_obs", "g_trans", "g_post", "d_a", "d_mean_smooth", "d_cov_smooth",
    "d_mat_As", "d_mat_Cs", "d_Q", "d_R", "d_m0", "d_P0", "g_scale",
    ("g_scale_stride", C.c_int32), ("_pad", C.c_int32),
    ("w_obs", C.c_double), ("w_trans", C.c_double), ("w_post", C.c_double)])
LossFwd = _ptr_struct("LossFwd", [
    "a", "alpha", "mask", "A_K", "C_K", "Q", "R", "m0", "P0", "eps", "saved", "terms", "out",
    ("w_obs", C.c_double), ("w_trans", C.c_double), ("w_post", C.c_double), "status"])
LossBwd = _ptr_struct("LossBwd", [
    ("fwd", LossFwd), "g_scale", ("g_scale_stride", C.c_int32), ("_pad", C.c_int32),
    "d_a", "d_alpha", "d_A_K", "d_C_K", "d_Q", "d_R", "d_m0", "d_P0"])
MaskedMlp = _ptr_struct("MaskedMlp", [
    ("hidden", C.c_int32), ("_pad", C.c_int32), "packed", "eps_z", "L_Q", "eps_a", "L_R", "a_w"])
MaskedMlpBwd = _ptr_struct("MaskedMlpBwd", [
    ("fwd", MaskedMlp), "g_a_w", "g_alpha", "d_logits", "d_L_Q", "d_L_R",
    ("learn_weight_model", C.c_int32), ("_pad", C.c_int32)])
Rollout = _ptr_struct("Rollout", [
    "A_K", "C_K", "Q", "mean_pred", "cov_pred", "mat_C", "eps", "packed_mlp", "alpha",
    ("hidden", C.c_int32), ("phase", C.c_int32), ("sample", C.c_int32), ("_pad", C.c_int32),
    "as_new", "means", "covs", "mean_pred_new", "cov_pred_new", "mat_As_new", "mat_Cs_new",
    "status"])
ROLLOUT_ALL, ROLLOUT_EMIT, ROLLOUT_ADVANCE = 0, 1, 2

EXPORTS = (
    "kvae_version", "kvae_error_string", "kvae_supported", "kvae_workspace_bytes",
    "kvae_filter_forward", "kvae_filter_backward",
    "kvae_filter_backward_noalloc", "kvae_noalloc_scratch_bytes",
    "kvae_smooth_forward", "kvae_smooth_backward", "kvae_filter_smooth_forward",
    "kvae_elbo_forward", "kvae_elbo_backward", "kvae_impute_observation",
    "kvae_kalman_loss_forward", "kvae_kalman_loss_backward", "kvae_kalman_loss_value_and_grad",
    "kvae_loss_saved_bytes",
    "kvae_loss_supported", "kvae_impute_observation_backward",
    "kvae_filter_forward_masked_mlp", "kvae_predict_future",
    "kvae_filter_smooth_fused", "kvae_filter_smooth_fused_workspace_bytes",
    "kvae_filter_backward_masked_mlp", "kvae_filter_backward_masked_mlp_workspace_bytes",
    "kvae_lstm_layer_forward", "kvae_lstm_layer_backward",
    "kvae_lstm2_forward", "kvae_lstm2_backward", "kvae_lstm2_backward_blocks",
    "kvae_lstm2_prepare", "kvae_lstm2_prepared_elems",
    "kvae_lstm128_blocks", "kvae_lstm128_forward", "kvae_lstm128_backward",
    "kvae_lstm128_backward_head", "kvae_lstm128_step_fused", "kvae_lstm128_cell_backward",
    "kvae_lstm128_set_precision", "kvae_lstm128_get_precision", "kvae_rowstream_matmul",
    "kvae_lstm128_wgrad_slabs", "kvae_lstm128_wgrad",
    "kvae_softmax_head_forward", "kvae_softmax_head_backward",
    "kvae_softmax_head_backward_blocks",
    "kvae_cholesky_forward", "kvae_cholesky_backward", "kvae_noise_factors",
    "kvae_graph_count_memset_nodes", "kvae_graph_count_kernel_nodes",
    "kvae_column_sum", "kvae_column_sum_many", "kvae_vector_sum", "kvae_vector_sum_scratch_elems",
    "kvae_weighted_terms", "kvae_weighted_terms_scratch_elems",
)


def ptr(t):
    """Device (or host) address of a contiguous tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor passed to the Kalman kernels must be contiguous")
    return t.data_ptr()


def fill(struct, **tensors):
    for k, v in tensors.items():
        if isinstance(v, int):
            setattr(struct, k, v)
        else:
            setattr(struct, k, ptr(v))
    return struct


def dtype_code(dtype):
    if dtype == torch.float32:
        return KVAE_F32
    if dtype == torch.float64:
        return KVAE_F64
    raise TypeError(f"Kalman kernels support float32 and float64, got {dtype}")


_lib = None


def declare(lib, prefix="kvae_", with_stream=True):
    """Set argtypes/restype for the six compute entry points of `lib`."""
    tail = [_vp] if with_stream else []
    ws = [_vp, C.c_size_t]
    spec = {
        "filter_forward": [C.POINTER(Dims), C.POINTER(FilterFwd)] + tail,
        "filter_backward": [C.POINTER(Dims), C.POINTER(FilterBwd)] + (ws + tail if with_stream else []),
        "smooth_forward": [C.POINTER(Dims), C.POINTER(SmoothFwd)] + tail,
        "smooth_backward": [C.POINTER(Dims), C.POINTER(SmoothBwd)] + (ws + tail if with_stream else []),
        "elbo_forward": [C.POINTER(Dims), C.POINTER(ElboFwd)] + tail,
        "elbo_backward": [C.POINTER(Dims), C.POINTER(ElboBwd)] + (ws + tail if with_stream else []),
    }
    if hasattr(lib, prefix + "kalman_loss_forward"):
        spec["kalman_loss_forward"] = ([C.POINTER(Dims), C.POINTER(LossFwd)]
                                       + (ws + tail if with_stream else []))
        spec["kalman_loss_backward"] = ([C.POINTER(Dims), C.POINTER(LossBwd)]
                                        + (ws + tail if with_stream else []))
    if hasattr(lib, prefix + "kalman_loss_value_and_grad"):
        spec["kalman_loss_value_and_grad"] = ([C.POINTER(Dims), C.POINTER(LossBwd)]
                                              + (ws + tail if with_stream else []))
    for name, argtypes in spec.items():
        fn = getattr(lib, prefix + name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if with_stream and hasattr(lib, prefix + "filter_smooth_forward"):
        fn = getattr(lib, prefix + "filter_smooth_forward")
        fn.argtypes = [C.POINTER(Dims), C.POINTER(FilterFwd), C.POINTER(SmoothFwd), _vp]
        fn.restype = C.c_int
    return lib


def load():
    """Load libkvae_kalman.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP library {LIB_PATH} not found: build it with "
            "`make -C kalman_vae_amd/csrc` (or __graft_entry__.build()). "
            "kalman_vae_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    declare(lib)
    lib.kvae_version.restype = C.c_int
    lib.kvae_error_string.restype = C.c_char_p
    lib.kvae_error_string.argtypes = [C.c_int]
    lib.kvae_supported.argtypes = [C.c_int] * 4
    lib.kvae_supported.restype = C.c_int
    lib.kvae_workspace_bytes.argtypes = [C.POINTER(Dims), C.c_int]
    lib.kvae_workspace_bytes.restype = C.c_size_t
    lib.kvae_loss_saved_bytes.argtypes = [C.POINTER(Dims)]
    lib.kvae_loss_saved_bytes.restype = C.c_size_t
    lib.kvae_loss_supported.argtypes = [C.c_int] * 4
    lib.kvae_loss_supported.restype = C.c_int
    lib.kvae_filter_backward_noalloc.argtypes = [C.POINTER(Dims), C.POINTER(FilterBwd), _vp]
    lib.kvae_filter_backward_noalloc.restype = C.c_int
    lib.kvae_noalloc_scratch_bytes.argtypes = []
    lib.kvae_noalloc_scratch_bytes.restype = C.c_size_t
    lib.kvae_lstm_layer_forward.argtypes = [C.c_int] * 3 + [_vp] * 5
    lib.kvae_lstm_layer_forward.restype = C.c_int
    lib.kvae_lstm_layer_backward.argtypes = [C.c_int] * 3 + [_vp] * 7
    lib.kvae_lstm_layer_backward.restype = C.c_int
    lib.kvae_lstm2_forward.argtypes = [C.c_int] * 3 + [_vp] * 16
    lib.kvae_lstm2_forward.restype = C.c_int
    lib.kvae_lstm2_backward.argtypes = [C.c_int] * 3 + [_vp] * 15
    lib.kvae_lstm2_backward_blocks.argtypes = [C.c_int]
    lib.kvae_lstm2_backward_blocks.restype = C.c_int
    lib.kvae_lstm2_backward.restype = C.c_int
    lib.kvae_lstm128_blocks.argtypes = [C.c_int]
    lib.kvae_lstm128_blocks.restype = C.c_int
    lib.kvae_lstm128_set_precision.argtypes = [C.c_int]
    lib.kvae_lstm128_set_precision.restype = C.c_int
    lib.kvae_lstm128_get_precision.argtypes = []
    lib.kvae_lstm128_get_precision.restype = C.c_int
    lib.kvae_rowstream_matmul.argtypes = [C.c_long, C.c_int, C.c_int] + [_vp] * 5
    lib.kvae_rowstream_matmul.restype = C.c_int
    lib.kvae_lstm128_wgrad_slabs.argtypes = [C.c_long]
    lib.kvae_lstm128_wgrad_slabs.restype = C.c_int
    lib.kvae_lstm128_wgrad.argtypes = [C.c_long, _vp, _vp, C.c_long, _vp, C.c_long, _vp]
    lib.kvae_lstm128_wgrad.restype = C.c_int
    lib.kvae_lstm128_forward.argtypes = [C.c_int] * 3 + [_vp] * 9 + [C.c_int, _vp]
    lib.kvae_lstm128_forward.restype = C.c_int
    lib.kvae_lstm128_backward.argtypes = [C.c_int] * 3 + [_vp] * 11
    lib.kvae_lstm128_backward.restype = C.c_int
    lib.kvae_lstm128_backward_head.argtypes = [C.c_int] * 3 + [_vp] * 11
    lib.kvae_lstm128_backward_head.restype = C.c_int
    lib.kvae_lstm128_step_fused.argtypes = [C.c_long, C.c_int] + [_vp] * 9
    lib.kvae_lstm128_step_fused.restype = C.c_int
    lib.kvae_lstm128_cell_backward.argtypes = [C.c_long] + [_vp] * 9 + [C.c_int] + [_vp] * 3
    lib.kvae_lstm128_cell_backward.restype = C.c_int
    lib.kvae_softmax_head_forward.argtypes = ([C.c_long, C.c_int, C.c_int] + [_vp] * 4
                                              + [C.c_long, _vp])
    lib.kvae_softmax_head_forward.restype = C.c_int
    lib.kvae_softmax_head_backward.argtypes = [C.c_long, C.c_int, C.c_int] + [_vp] * 7
    lib.kvae_softmax_head_backward.restype = C.c_int
    lib.kvae_softmax_head_backward_blocks.argtypes = [C.c_long]
    lib.kvae_softmax_head_backward_blocks.restype = C.c_int
    lib.kvae_impute_observation.argtypes = [C.POINTER(Dims)] + [_vp] * 12
    lib.kvae_impute_observation.restype = C.c_int
    lib.kvae_impute_observation_backward.argtypes = [C.POINTER(Dims)] + [_vp] * 18
    lib.kvae_impute_observation_backward.restype = C.c_int
    lib.kvae_filter_forward_masked_mlp.argtypes = [C.POINTER(Dims), C.POINTER(FilterFwd),
                                                   C.POINTER(MaskedMlp), _vp]
    lib.kvae_filter_forward_masked_mlp.restype = C.c_int
    lib.kvae_predict_future.argtypes = [C.POINTER(Dims), C.POINTER(Rollout), _vp]
    lib.kvae_predict_future.restype = C.c_int
    lib.kvae_filter_smooth_fused_workspace_bytes.argtypes = [C.POINTER(Dims), C.POINTER(FilterFwd)]
    lib.kvae_filter_smooth_fused_workspace_bytes.restype = C.c_size_t
    lib.kvae_filter_smooth_fused.argtypes = [C.POINTER(Dims), C.POINTER(FilterFwd),
                                             C.POINTER(MaskedMlp), C.POINTER(SmoothFwd), _vp,
                                             C.c_size_t, _vp]
    lib.kvae_filter_smooth_fused.restype = C.c_int
    lib.kvae_filter_backward_masked_mlp_workspace_bytes.argtypes = [C.POINTER(Dims)]
    lib.kvae_filter_backward_masked_mlp_workspace_bytes.restype = C.c_size_t
    lib.kvae_filter_backward_masked_mlp.argtypes = [C.POINTER(Dims), C.POINTER(FilterBwd),
                                                    C.POINTER(MaskedMlpBwd), _vp, C.c_size_t, _vp]
    lib.kvae_filter_backward_masked_mlp.restype = C.c_int
    lib.kvae_lstm2_prepare.argtypes = [_vp] * 9
    lib.kvae_lstm2_prepare.restype = C.c_int
    lib.kvae_lstm2_prepared_elems.argtypes = []
    lib.kvae_lstm2_prepared_elems.restype = C.c_long
    lib.kvae_cholesky_forward.argtypes = [C.c_int, C.c_int] + [_vp] * 4
    lib.kvae_cholesky_forward.restype = C.c_int
    lib.kvae_cholesky_backward.argtypes = [C.c_int, C.c_int] + [_vp] * 4
    lib.kvae_cholesky_backward.restype = C.c_int
    lib.kvae_noise_factors.argtypes = ([C.c_int] * 3 + [_vp, _vp, C.c_double, C.c_double]
                                       + [_vp] * 5)
    lib.kvae_noise_factors.restype = C.c_int
    for name in ("kvae_graph_count_memset_nodes", "kvae_graph_count_kernel_nodes"):
        getattr(lib, name).argtypes = [_vp]
        getattr(lib, name).restype = C.c_int
    lib.kvae_weighted_terms.argtypes = [C.c_long, C.c_int, _vp, _vp, C.c_double, C.c_double,
                                        C.c_double, _vp, _vp]
    lib.kvae_weighted_terms.restype = C.c_int
    lib.kvae_weighted_terms_scratch_elems.argtypes = [C.c_long]
    lib.kvae_weighted_terms_scratch_elems.restype = C.c_long
    lib.kvae_column_sum.argtypes = [C.c_long, C.c_int, C.c_int, _vp, _vp, _vp]
    lib.kvae_column_sum.restype = C.c_int
    lib.kvae_column_sum_many.argtypes = [C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_int,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp]
    lib.kvae_column_sum_many.restype = C.c_int
    lib.kvae_vector_sum.argtypes = [C.c_int, C.c_long, C.c_int, _vp, _vp, _vp, _vp]
    lib.kvae_vector_sum.restype = C.c_int
    lib.kvae_vector_sum_scratch_elems.argtypes = [C.c_int, C.c_long]
    lib.kvae_vector_sum_scratch_elems.restype = C.c_long
    _lib = lib
    return lib


def column_sum(x):
    """Column sums of a contiguous 2-D GPU tensor (deterministic, no autograd)."""
    require_gpu(x)
    out = torch.empty(x.shape[1], dtype=x.dtype, device=x.device)
    check(load().kvae_column_sum(x.shape[0], x.shape[1], dtype_code(x.dtype), ptr(x),
                                 out.data_ptr(), current_stream_ptr()))
    return out


def column_sum_many(xs):
    """Column sums of up to four contiguous 2-D GPU tensors of one dtype in one launch."""
    require_gpu(*xs)
    n = len(xs)
    outs = [torch.empty(x.shape[1], dtype=x.dtype, device=x.device) for x in xs]
    rows = (C.c_long * n)(*[x.shape[0] for x in xs])
    cols = (C.c_int * n)(*[x.shape[1] for x in xs])
    ins = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    ous = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    check(load().kvae_column_sum_many(n, rows, cols, dtype_code(xs[0].dtype), ins, ous,
                                      current_stream_ptr()))
    return outs


def column_sum_tall(x):
    """Column sums of a contiguous (rows, cols) GPU tensor with many rows and few
    columns: kvae_column_sum gives every 64 columns one workgroup, so k consecutive
    rows are first viewed as one row of k*cols columns (k workgroup groups share the
    rows), then the k partial sums per column are added by a second call."""
    rows, cols = x.shape
    k = 1
    while k < 256 and rows % (2 * k) == 0 and rows // (2 * k) >= 64:
        k *= 2
    if k == 1:
        return column_sum(x)
    return column_sum(column_sum(x.view(rows // k, k * cols)).view(k, cols))


def check(code):
    if code != 0:
        raise RuntimeError("kvae kernel call failed: "
                           + load().kvae_error_string(code).decode())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "kalman_vae_amd runs on MI355X only: got a tensor on "
                f"{t.device}; there is no CPU fallback")


def current_stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def status_message(word):
    return "; ".join(msg for bit, msg in STATUS_BITS.items() if word & bit)
