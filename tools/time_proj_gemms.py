#!/usr/bin/env python3
"""The two activation-side GEMMs of the LSTM weight network at the benchmark size
(81920 x 128 x 512 and 81920 x 512 x 128) in several operand layouts, HIP events."""
import torch


def timed(fn, iters=30):
    for _ in range(5):
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
    dev = torch.device("cuda:0")
    M, H, G = 81920, 128, 512
    h = torch.randn(M, H, device=dev)
    W = torch.randn(G, H, device=dev) * 0.1          # (4H, H) as torch.nn.LSTM stores it
    Wt = W.t().contiguous()                           # (H, 4H)
    b = torch.randn(G, device=dev)
    out = torch.empty(M, G, device=dev)
    dG = torch.randn(M, G, device=dev)
    gh = torch.empty(M, H, device=dev)
    fl = 2.0 * M * H * G
    for name, fn in [
        ("addmm(b, h, W.t())", lambda: torch.addmm(b, h, W.t(), out=out)),
        ("addmm(b, h, Wt)", lambda: torch.addmm(b, h, Wt, out=out)),
        ("mm(h, Wt)", lambda: torch.mm(h, Wt, out=out)),
        ("mm(h, W.t())", lambda: torch.mm(h, W.t(), out=out)),
        ("linear(h, W, b)", lambda: torch.nn.functional.linear(h, W, b)),
        ("mm(dG, W)", lambda: torch.mm(dG, W, out=gh)),
        ("mm(dG, Wt.t())", lambda: torch.mm(dG, Wt.t(), out=gh)),
    ]:
        us = timed(fn)
        print(f"{name:24s} {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s", flush=True)
    for pref in ("hipblaslt", "hipblas"):
        try:
            torch.backends.cuda.preferred_blas_library(pref)
        except Exception as e:
            print(pref, "unavailable:", e)
            continue
        us1 = timed(lambda: torch.addmm(b, h, W.t(), out=out))
        us2 = timed(lambda: torch.mm(dG, W, out=gh))
        print(f"preferred {pref}: addmm {us1:.1f} us, mm(dG, W) {us2:.1f} us", flush=True)


if __name__ == "__main__":
    main()
