#!/usr/bin/env python3
"""One time step of the weight network's LSTM over a chunk of the masked-inference path
(262144 rows): the launches of the step, HIP-event durations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kalman_vae_amd import _capi as capi   # noqa: E402


def timed(fn, iters=10):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    H, P = 128, 2
    dev = torch.device("cuda:0")
    lib = capi.load()
    new = lambda *s: torch.randn(*s, device=dev)
    x, wih0, b0 = new(1, B, P), 0.1 * new(4 * H, P), 0.1 * new(4 * H)
    whh0t, wih1t, whh1t, b1 = 0.08 * new(H, 4 * H), 0.08 * new(H, 4 * H), 0.08 * new(H, 4 * H), 0.1 * new(4 * H)
    h0p, c0p, h1p, c1p = (0.5 * new(B, H) for _ in range(4))
    h0, c0, h1, c1 = (new(B, H) for _ in range(4))
    gates = new(B, 4 * H)
    st = torch.cuda.current_stream().cuda_stream
    t_l0 = timed(lambda: lib.kvae_lstm128_forward(1, B, P, x.data_ptr(), wih0.data_ptr(), b0.data_ptr(),
                                                  whh0t.data_ptr(), None, h0.data_ptr(), c0.data_ptr(),
                                                  h0p.data_ptr(), c0p.data_ptr(), 0, st))
    t_proj = timed(lambda: lib.kvae_rowstream_matmul(B, H, 4 * H, h0.data_ptr(), wih1t.data_ptr(),
                                                     b1.data_ptr(), gates.data_ptr(), st))
    t_l1 = timed(lambda: lib.kvae_lstm128_forward(1, B, 0, None, None, None, whh1t.data_ptr(),
                                                  gates.data_ptr(), h1.data_ptr(), c1.data_ptr(),
                                                  h1p.data_ptr(), c1p.data_ptr(), 0, st))
    t_f = timed(lambda: lib.kvae_lstm128_step_fused(B, 0, h0.data_ptr(), wih1t.data_ptr(), h1p.data_ptr(),
                                                    c1p.data_ptr(), whh1t.data_ptr(), b1.data_ptr(),
                                                    h1.data_ptr(), c1.data_ptr(), st))
    t_f0 = timed(lambda: lib.kvae_lstm128_step_fused(B, P, x.data_ptr(), wih0.data_ptr(), h0p.data_ptr(),
                                                     c0p.data_ptr(), whh0t.data_ptr(), b0.data_ptr(),
                                                     h0.data_ptr(), c0.data_ptr(), st))
    kb = lambda per_row: per_row * B / 1e6
    print(f"B={B}: layer 0 step: many-tiles recurrent kernel {t_l0:.0f} us ({kb(2056) / t_l0 * 1e3:.0f} "
          f"GB/s), step kernel {t_f0:.0f} us ({kb(2056) / t_f0 * 1e3:.0f} GB/s); layer 1: projection "
          f"{t_proj:.0f} us + recurrent step {t_l1:.0f} us = {t_proj + t_l1:.0f} us "
          f"({kb(2560 + 4096) / (t_proj + t_l1) * 1e3:.0f} GB/s), step kernel {t_f:.0f} us "
          f"({kb(3584) / t_f * 1e3:.0f} GB/s incl. the second read of x and h_prev)")


if __name__ == "__main__":
    main()
