"""Variants of the LSTM weight-gradient product dW (512,128) = dG^T (512,R) h (R,128), R = T*B."""
import sys
import torch

R = int(sys.argv[1]) if len(sys.argv) > 1 else 81920
dev = torch.device("cuda:0")
d = torch.randn(R, 512, device=dev)
h = torch.randn(R, 128, device=dev)
h2 = torch.randn(R, 256, device=dev)


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


ref = (d.double().t() @ h.double()).float()
for S in (10, 16, 20, 32, 40, 64, 80):
    if R % S:
        continue
    c = R // S
    f1 = lambda: torch.bmm(d.view(S, c, 512).transpose(1, 2), h.view(S, c, 128))
    f2 = lambda: torch.bmm(h.view(S, c, 128).transpose(1, 2), d.view(S, c, 512))
    f3 = lambda: torch.bmm(d.view(S, c, 512).transpose(1, 2), h2.view(S, c, 256))
    err = float((f1().sum(0) - ref).abs().max() / ref.abs().max())
    print(f"S={S:3d} chunk={c:6d}: dG^T h {timed(f1):7.1f} us | h^T dG {timed(f2):7.1f} us | "
          f"dG^T [h0|h1] (N=256) {timed(f3):7.1f} us (={timed(f3)/2:6.1f} per matrix)  err {err:.1e}")
