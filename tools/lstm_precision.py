#!/usr/bin/env python3
"""Error of the register-resident LSTM layer kernels against a float64 torch.nn.LSTM for
both product forms (exact fp32 matrix instructions / three bf16 instructions per product),
and their HIP-event durations.  T=20, H=128, two layers, inputs of the bench's size."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kalman_vae_amd import _capi as capi   # noqa: E402
import kalman_vae_amd.lstm_function as LF  # noqa: E402


def rel(x, y):
    return float((x.double() - y).abs().max() / y.abs().max())


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
    lib = capi.load()
    for seed, scale in ((0, 1.0), (1, 3.0)):
        torch.manual_seed(seed)
        lstm = torch.nn.LSTM(P, H, num_layers=2).to(dev)
        with torch.no_grad():
            for p_ in lstm.parameters():
                p_.mul_(scale)
        x = torch.randn(T, B, P, device=dev, requires_grad=True)
        w = torch.randn(T, B, H, device=dev)
        ref = torch.nn.LSTM(P, H, num_layers=2).to(dev).double()
        ref.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
        x64 = x.detach().double().requires_grad_(True)
        with torch.backends.cudnn.flags(enabled=False):
            o64, _ = ref(x64)
            g64 = torch.autograd.grad((o64 * w.double()).sum(), [x64] + list(ref.parameters()))
            o32, _ = lstm(x)
            g32 = torch.autograd.grad((o32 * w).sum(), [x] + list(lstm.parameters()))
        names = ["x"] + [n for n, _ in lstm.named_parameters()]
        print(f"weights x{scale}: torch fp32 (library): out {rel(o32, o64):.2e}, worst gradient "
              f"{max(rel(a, b) for a, b in zip(g32, g64)):.2e}", flush=True)
        for split in (0, 1):
            assert lib.kvae_lstm128_set_precision(split) == 0
            out, _ = LF.fused_lstm(x, lstm)
            g = torch.autograd.grad((out * w).sum(), [x] + list(lstm.parameters()))
            errs = {n: rel(a, b) for n, a, b in zip(names, g, g64)}
            worst = max(errs, key=errs.get)
            print(f"weights x{scale}: split={split}: out {rel(out, o64):.2e}, worst gradient "
                  f"{errs[worst]:.2e} ({worst})", flush=True)
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(P, H, num_layers=2).to(dev)
    x = torch.randn(T, B, P, device=dev, requires_grad=True)
    w = torch.randn(T, B, H, device=dev) * 1e-3
    new = lambda *s: torch.empty(s, device=dev)
    gates, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    whht = lstm.weight_hh_l0.detach().t().contiguous()
    whh = lstm.weight_hh_l0.detach().contiguous()
    wih = lstm.weight_ih_l0.detach().contiguous()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach()
    xd = x.detach()
    st = torch.cuda.current_stream().cuda_stream
    gx = torch.randn(T, B, 4 * H, device=dev)
    dG, dx = new(T, B, 4 * H), new(T, B, P)
    part = new(lib.kvae_lstm128_blocks(B), 4 * H * (2 + P))
    flops_layer = 2.0 * T * B * 4 * H * H
    for split in (0, 1):
        lib.kvae_lstm128_set_precision(split)
        t0 = timed(lambda: lib.kvae_lstm128_forward(T, B, P, xd.data_ptr(), wih.data_ptr(),
                                                    bias.data_ptr(), whht.data_ptr(), gates.data_ptr(),
                                                    h.data_ptr(), c.data_ptr(), None, None, 1, st))
        t1 = timed(lambda: lib.kvae_lstm128_forward(T, B, 0, None, None, None, whht.data_ptr(),
                                                    gx.data_ptr(), h.data_ptr(), c.data_ptr(), None,
                                                    None, 1, st))
        t1n = timed(lambda: lib.kvae_lstm128_forward(T, B, 0, None, None, None, whht.data_ptr(),
                                                     gx.data_ptr(), h.data_ptr(), c.data_ptr(), None,
                                                     None, 0, st))
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
        print(f"split={split}: forward P=2 {t0:.1f} us, forward gates given {t1:.1f} us "
              f"(no gate stores {t1n:.1f}), backward P=2 {tb0:.1f} us, backward P=0 {tb1:.1f} us; "
              f"{flops_layer / 1e9:.1f} GFLOP per launch", flush=True)
    lib.kvae_lstm128_set_precision(1)


if __name__ == "__main__":
    main()
