#!/usr/bin/env python3
"""Forward + backward of the default weight network (LSTM 2x128 + softmax head) at the bench
shape: the head as its own autograd node against the head's adjoint inside the upper layer's
backward kernel, for K = 2, 3, 4."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    T, P = 20, 2
    dev = torch.device("cuda:0")
    for K in (2, 3, 4):
        torch.manual_seed(K)
        lstm = torch.nn.LSTM(P, 128, num_layers=2).to(dev)
        lin = torch.nn.Linear(128, K).to(dev)
        x = torch.randn(T, B, P, device=dev, requires_grad=True)
        w = torch.randn(T + 1, B, K, device=dev)
        params = [x] + list(lstm.parameters()) + list(lin.parameters())

        def separate():
            h, _ = LF.fused_lstm(x, lstm)
            torch.autograd.grad((LF.softmax_head(h, lin, True) * w).sum(), params)

        def fused():
            a, _ = LF.fused_lstm_head(x, lstm, lin, True)
            torch.autograd.grad((a * w).sum(), params)

        print(f"K={K}: separate nodes {timed(separate):.0f} us, one node {timed(fused):.0f} us "
              "(eager, launch overhead included)", flush=True)


if __name__ == "__main__":
    main()
