"""kvae_filter_backward_noalloc (the entry point for callers that cannot allocate: its
scratch is a fixed array inside the library) against kvae_filter_backward with a
caller-provided workspace, through the C ABI: bit-identical outputs on two calls in a
row, the second on a smaller problem (so the array still holds the first call's rows),
and a clear error beyond the array's size."""
import ctypes as C

import pytest
import torch

from kalman_vae_amd import _capi as capi
from kernel_harness import Backend, random_problem

pytestmark = pytest.mark.gpu

OUTS = ("d_a", "d_alpha", "d_m_in", "d_P_in", "d_A_K", "d_C_K", "d_Q", "d_R")


def _run_both(be, T, B, n, p, K, dtype, seed, burn_in):
    prob = random_problem(T, B, n, p, K, dtype, seed, mask_kind="binary")
    dv = {k: (v.to(be.device).contiguous() if v is not None else None) for k, v in prob.items()}
    fwd = be.filter_forward(dv["a"], dv["alpha"], dv["mask"], dv["A_K"], dv["C_K"], dv["Q"],
                            dv["R"], dv["m0"], dv["P0"])
    g = torch.Generator().manual_seed(seed + 5)
    grads = {k: torch.randn(fwd[k].shape, generator=g, dtype=torch.float64).to(dtype)
             .to(be.device).contiguous()
             for k in ("mean_filt", "cov_filt", "mean_pred", "cov_pred", "mat_As", "mat_Cs")}
    want = be.filter_backward(fwd, grads, burn_in=burn_in)      # workspace from the caller

    d = be.dims(T, B, n, p, K, dtype, True, burn_in)
    got = {k: torch.full_like(v, float("nan")) for k, v in want.items()}
    io = capi.FilterBwd()
    io.fwd = fwd["_io"]
    capi.fill(io, **{"g_" + k: v for k, v in grads.items()}, **got)
    lib = capi.load()
    before = torch.cuda.memory_allocated()
    code = lib.kvae_filter_backward_noalloc(C.byref(d), C.byref(io),
                                            torch.cuda.current_stream().cuda_stream)
    assert code == 0, lib.kvae_error_string(code)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    for k in OUTS:
        assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_noalloc_filter_backward_matches_workspace_call(dtype):
    be = Backend("hip")
    # two calls in a row on the same GPU: the second is smaller and leaves most of the
    # rows the first one wrote in place
    _run_both(be, 9, 1000, 4, 2, 3, dtype, seed=3, burn_in=2)
    _run_both(be, 5, 70, 4, 2, 3, dtype, seed=4, burn_in=0)
    # a large-state shape small enough for the array, then the small state again
    _run_both(be, 4, 3, 16, 8, 4, dtype, seed=6, burn_in=1)
    _run_both(be, 6, 33, 3, 2, 2, dtype, seed=7, burn_in=0)


def test_noalloc_rejects_sizes_beyond_its_scratch():
    lib = capi.load()
    limit = lib.kvae_noalloc_scratch_bytes()
    assert limit == 4 << 20
    # the large-state workspace grows with T*B*n*n: far beyond 4 MiB here
    d = capi.Dims(200, 512, 32, 16, 8, capi.KVAE_F32, 1, 0)
    assert lib.kvae_workspace_bytes(C.byref(d), capi.OP_FILTER_BWD) > limit
    io = capi.FilterBwd()
    code = lib.kvae_filter_backward_noalloc(C.byref(d), C.byref(io), None)
    assert code == -5
    assert b"no-allocation" in lib.kvae_error_string(code)
