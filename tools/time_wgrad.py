#!/usr/bin/env python3
"""HIP-event durations of the LSTM weight-gradient products at the bench shape: the
library's slab-batched fp32 GEMMs against kvae_lstm128_wgrad."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kalman_vae_amd import _capi as capi   # noqa: E402
import kalman_vae_amd.lstm_function as LF  # noqa: E402


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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T, H = 20, 128
    R, E = T * B, 4 * H * H
    dev = torch.device("cuda:0")
    lib = capi.load()
    d0, d1 = torch.randn(R, 4 * H, device=dev), torch.randn(R, 4 * H, device=dev)
    h0, h1 = torch.randn(R, H, device=dev), torch.randn(R, H, device=dev)
    t_lib = timed(lambda: LF._splitk_tn_many([(d0[B:], h0[:-B]), (d1, h0), (d1[B:], h1[:-B])]))
    S = lib.kvae_lstm128_wgrad_slabs(R)
    slabs = torch.empty(S, 3 * E, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    t1 = timed(lambda: lib.kvae_lstm128_wgrad(R, d0.data_ptr(), h0.data_ptr(), B,
                                              slabs.data_ptr(), 3 * E, st))
    t2 = timed(lambda: lib.kvae_lstm128_wgrad(R, d1.data_ptr(), h0.data_ptr(), 0,
                                              slabs[:, E:].data_ptr(), 3 * E, st))
    t3 = timed(lambda: capi.column_sum(slabs))
    by1 = 4.0 * R * (4 * H + H)
    print(f"library (3 slab-batched GEMMs + slab sum): {t_lib:.1f} us")
    print(f"kvae_lstm128_wgrad: with row offset {t1:.1f} us ({by1 / t1 / 1e3:.0f} GB/s), without "
          f"{t2:.1f} us ({by1 / t2 / 1e3:.0f} GB/s), slab sum of three {t3:.1f} us; "
          f"three products + sum {t1 + 2 * t2 + t3:.1f} us")


if __name__ == "__main__":
    main()
