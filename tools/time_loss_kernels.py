#!/usr/bin/env python3
"""HIP-event durations of the fused objective's two calls and of the six separate kernels
on the same problem.  python tools/time_loss_kernels.py [B ...]   (KVAE_TIME_T=<steps>, default 20)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kalman_vae_amd import _capi as capi          # noqa: E402
from kernel_harness import Backend, random_problem  # noqa: E402


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    T, n, p, K = int(os.environ.get("KVAE_TIME_T", "20")), 4, 2, 3
    dt = torch.float32
    lib = capi.load()
    be = Backend("hip")
    for B in [int(x) for x in sys.argv[1:]] or [4096, 16384, 65536]:
        prob = random_problem(T, B, n, p, K, dt, 0)
        dv = {k: (v.cuda().contiguous() if v is not None else None) for k, v in prob.items()}
        fwd = be.loss_forward(dv["a"], dv["alpha"], None, dv["A_K"], dv["C_K"], dv["Q"], dv["R"],
                              dv["m0"], dv["P0"], dv["eps_e"], (1.0, 1.0, -1.0), want_terms=False)
        d = be.dims(T, B, n, p, K, dt, True, 3)
        stream = torch.cuda.current_stream().cuda_stream
        nb_f = lib.kvae_workspace_bytes(C.byref(d), capi.OP_LOSS_FWD)
        nb_b = lib.kvae_workspace_bytes(C.byref(d), capi.OP_LOSS_BWD)
        ws = torch.empty(max(nb_f, nb_b), dtype=torch.uint8, device="cuda")
        g_scale = torch.ones(1, dtype=dt, device="cuda")
        bwd = be.loss_backward(fwd, g_scale, 0, burn_in=3)
        io_b = capi.LossBwd()
        io_b.fwd = fwd["_io"]
        capi.fill(io_b, g_scale=g_scale, g_scale_stride=0,
                  **{k: v for k, v in bwd.items() if k.startswith("d_")})
        tf = timed(lambda: lib.kvae_kalman_loss_forward(C.byref(d), C.byref(fwd["_io"]),
                                                        ws.data_ptr(), nb_f, stream))
        tb = timed(lambda: lib.kvae_kalman_loss_backward(C.byref(d), C.byref(io_b),
                                                         ws.data_ptr(), nb_b, stream))
        nb_v = lib.kvae_workspace_bytes(C.byref(d), capi.OP_LOSS_VALUE_AND_GRAD)
        ws_v = torch.empty(nb_v, dtype=torch.uint8, device="cuda")
        tv = timed(lambda: lib.kvae_kalman_loss_value_and_grad(C.byref(d), C.byref(io_b),
                                                               ws_v.data_ptr(), nb_v, stream))
        units = T * B
        flops = 5.6e3 * units
        print(f"T={T} B={B} lanes={os.environ.get('KVAE_SEQ_LANES', 'auto')}: value_and_grad {tv:.1f} us"
              f" = {units / tv * 1e6 / 1e9:.2f} G seq-steps/s,"
              f" {flops / tv / 1e6 / 157.3 * 100:.1f} % of fp32 VALU peak", flush=True)
        print(f"B={B}: loss_forward {tf:.1f} us, loss_backward {tb:.1f} us, total {tf + tb:.1f} us"
              f" = {units / (tf + tb) * 1e6 / 1e9:.2f} G seq-steps/s,"
              f" {flops / (tf + tb) / 1e6 / 157.3 * 100:.1f} % of fp32 VALU peak", flush=True)


if __name__ == "__main__":
    main()
