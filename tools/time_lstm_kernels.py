"""Time the persistent two-layer LSTM kernels through the C-ABI for both tile
heights, and the weight-gradient GEMM variants around them.
usage: python tools/time_lstm_kernels.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalman_vae_amd import _capi as capi  # noqa: E402

if os.environ.get("KVAE_ALT_LIB"):       # time an experimental build of the library
    capi.LIB_PATH = os.environ["KVAE_ALT_LIB"]
dev = torch.device("cuda:0")
T, P, H = 20, 2, 128


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def lstm_kernels(B):
    lib = capi.load()
    torch.manual_seed(0)
    x = torch.randn(T, B, P, device=dev)
    w = lambda *s: torch.randn(*s, device=dev) * 0.08
    wih0, b0, whh0, wih1, whh1, b1 = w(4 * H, P), w(4 * H), w(4 * H, H), w(4 * H, H), w(4 * H, H), w(4 * H)
    new = lambda *s: torch.empty(*s, device=dev)
    g0, h0, c0, g1, h1, c1 = new(T, B, 4 * H), new(T, B, H), new(T, B, H), new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    dG0, dG1 = torch.empty_like(g0), torch.empty_like(g1)
    gh = torch.randn(T, B, H, device=dev)
    whh0T, wih1T, whh1T = whh0.t().contiguous(), wih1.t().contiguous(), whh1.t().contiguous()
    st = capi.current_stream_ptr()

    def fwd():
        capi.check(lib.kvae_lstm2_forward(T, B, P, x.data_ptr(), wih0.data_ptr(), b0.data_ptr(),
                                          whh0T.data_ptr(), wih1T.data_ptr(), whh1T.data_ptr(),
                                          b1.data_ptr(), g0.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                          g1.data_ptr(), h1.data_ptr(), c1.data_ptr(), None, None, st))

    dx = torch.empty_like(x)
    part = torch.empty(lib.kvae_lstm2_backward_blocks(B), 4 * H * (4 + P), device=dev)

    def bwd():
        capi.check(lib.kvae_lstm2_backward(T, B, P, gh.data_ptr(), x.data_ptr(), g0.data_ptr(),
                                           c0.data_ptr(), g1.data_ptr(), c1.data_ptr(),
                                           wih0.data_ptr(), whh0.data_ptr(), wih1.data_ptr(),
                                           whh1.data_ptr(), dG0.data_ptr(), dG1.data_ptr(),
                                           dx.data_ptr(), part.data_ptr(), st))
    flops = 2.0 * T * B * 4 * H * (H + 2 * H)      # recurrent + layer-1 input products
    for rows in ((16,) if os.environ.get("KVAE_ONLY16") else (16, 32)):
        os.environ["KVAE_LSTM_TILE_ROWS"] = str(rows)
        tf, tb = timed(fwd), timed(bwd)
        print(f"B={B} rows={rows}: fwd {tf:8.1f} us ({flops / tf * 1e-6:6.1f} TFLOP/s)  "
              f"bwd {tb:8.1f} us ({flops / tb * 1e-6:6.1f} TFLOP/s)")
    os.environ.pop("KVAE_LSTM_TILE_ROWS")
    return dG1.view(T * B, 4 * H), h0.view(T * B, H)


def gemm_variants(d, h):
    R = d.shape[0]
    print(f"weight-gradient GEMM (512 x {R}) x ({R} x 128):")
    print(f"  plain a.t() @ b        {timed(lambda: d.t() @ h):8.1f} us")
    for chunk in (512, 1024, 2048, 4096, 8192):
        if R % chunk:
            continue
        S = R // chunk
        f = lambda: torch.bmm(d.view(S, chunk, -1).transpose(1, 2), h.view(S, chunk, -1)).sum(0)
        print(f"  split chunk={chunk:5d} S={S:4d} {timed(f):8.1f} us")


if __name__ == "__main__":
    Bs = [int(b) for b in sys.argv[1:]] or [4096, 16384]
    for B in Bs:
        d, h = lstm_kernels(B)
        if B == Bs[0]:
            gemm_variants(d, h)
