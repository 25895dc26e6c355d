#!/usr/bin/env python3
"""Times the experiment builds of the weight-gradient kernel (tools/exp/build_wgrad_variants.sh)."""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    B, T, H = 4096, 20, 128
    R, E = T * B, 4 * H * H
    dev = torch.device("cuda:0")
    d0, h0, h1 = (torch.randn(R, 4 * H, device=dev), torch.randn(R, H, device=dev),
                  torch.randn(R, H, device=dev))
    st = torch.cuda.current_stream().cuda_stream
    vp = C.c_void_p
    for path in sorted(glob.glob(os.path.join(ROOT, "tools/exp/_build/*.so"))):
        lib = C.CDLL(path)
        lib.kvae_lstm128_wgrad.argtypes = [C.c_long, vp, vp, C.c_long, vp, C.c_long, vp]
        lib.kvae_lstm128_wgrad_slabs.argtypes = [C.c_long]
        S = lib.kvae_lstm128_wgrad_slabs(R)
        slabs = torch.empty(S, 2 * E, device=dev)
        t1 = timed(lambda: lib.kvae_lstm128_wgrad(R, d0.data_ptr(), h0.data_ptr(), B,
                                                  slabs.data_ptr(), 2 * E, st))
        t2 = timed(lambda: lib.kvae_lstm128_wgrad(R, d0.data_ptr(), h1.data_ptr(), 0,
                                                  slabs.data_ptr(), 2 * E, st))
        print(f"{os.path.basename(path):28s} offset {t1:7.1f} us   plain {t2:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
