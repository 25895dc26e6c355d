#!/usr/bin/env python3
"""HIP-event durations of the weight network's LSTM at the bench shape: the one-launch
two-layer kernels (weights streamed from L2) against the one-launch-per-layer kernels with
register-resident recurrent matrices, forward and backward, piece by piece."""
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
    T, P, H = 20, 2, 128
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(P, H, num_layers=2).to(dev)
    x = torch.randn(T, B, P, device=dev, requires_grad=True)
    w = torch.randn(T, B, H, device=dev) * 1e-3
    lib = capi.load()
    flops_layer = 2.0 * T * B * 4 * H * H
    for name, resident in (("streamed 2-layer", False), ("resident per layer", True)):
        LF.RESIDENT = resident
        out, _ = LF.fused_lstm(x, lstm)
        tf = timed(lambda: LF.fused_lstm(x, lstm))

        def fb():
            o, _ = LF.fused_lstm(x, lstm)
            torch.autograd.grad((o * w).sum(), [x] + list(lstm.parameters()))
        tfb = timed(fb)
        print(f"{name}: forward {tf:.0f} us, forward+backward (incl. weight-gradient GEMMs) "
              f"{tfb:.0f} us", flush=True)
    # the resident kernels alone
    new = lambda *s: torch.empty(s, device=dev)
    gates, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    whht = lstm.weight_hh_l0.detach().t().contiguous()
    whh = lstm.weight_hh_l0.detach().contiguous()
    wih = lstm.weight_ih_l0.detach().contiguous()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach()
    xd = x.detach()
    st = torch.cuda.current_stream().cuda_stream
    gx = torch.randn(T, B, 4 * H, device=dev)
    t0 = timed(lambda: lib.kvae_lstm128_forward(T, B, P, xd.data_ptr(), wih.data_ptr(),
                                                bias.data_ptr(), whht.data_ptr(), gates.data_ptr(),
                                                h.data_ptr(), c.data_ptr(), None, None, 1, st))
    t1 = timed(lambda: lib.kvae_lstm128_forward(T, B, 0, None, None, None, whht.data_ptr(),
                                                gx.data_ptr(), h.data_ptr(), c.data_ptr(), None,
                                                None, 1, st))
    dG, dx = new(T, B, 4 * H), new(T, B, P)
    part = new(lib.kvae_lstm128_blocks(B), 4 * H * (2 + P))
    lib.kvae_lstm128_forward(T, B, P, xd.data_ptr(), wih.data_ptr(), bias.data_ptr(),
                             whht.data_ptr(), gates.data_ptr(), h.data_ptr(), c.data_ptr(), None,
                             None, 1, st)
    tb0 = timed(lambda: lib.kvae_lstm128_backward(T, B, P, xd.data_ptr(), wih.data_ptr(),
                                                  gates.data_ptr(), c.data_ptr(), None, w.data_ptr(),
                                                  whh.data_ptr(), dG.data_ptr(), dx.data_ptr(),
                                                  part.data_ptr(), st))
    tb1 = timed(lambda: lib.kvae_lstm128_backward(T, B, 0, None, None, gates.data_ptr(),
                                                  c.data_ptr(), None, w.data_ptr(), whh.data_ptr(),
                                                  dG.data_ptr(), None, part.data_ptr(), st))
    h2 = h.view(T * B, H)
    wih1t = torch.randn(H, 4 * H, device=dev)
    g2 = gx.view(T * B, 4 * H)
    tg = timed(lambda: torch.addmm(bias, h2, wih1t, out=g2))
    wih1 = torch.randn(4 * H, H, device=dev)
    tg2 = timed(lambda: torch.mm(g2, wih1))
    for nm, t in (("layer forward (x projected in kernel)", t0), ("layer forward (gates given)", t1),
                  ("layer backward (P=2)", tb0), ("layer backward (P=0)", tb1)):
        print(f"  {nm}: {t:.1f} us = {flops_layer / t / 1e6:.1f} TFLOP/s "
              f"({flops_layer / t / 1e6 / 157.3 * 100:.0f} % of the fp32 matrix peak)")
    print(f"  layer-1 input GEMM {tg:.1f} us ({flops_layer / tg / 1e6:.1f} TFLOP/s), "
          f"dG1 x Wih1 GEMM {tg2:.1f} us")


if __name__ == "__main__":
    main()
