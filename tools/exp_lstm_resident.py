#!/usr/bin/env python3
"""Where the register-resident LSTM forward kernel spends its time: the same kernel built
without its global stores, without the gate activations, without the per-step barrier
(tools/exp/liblstm_*.so, built with -DKVAE_EXP_*; results of those builds are wrong on
purpose).  python tools/exp_lstm_resident.py"""
import ctypes as C
import glob
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, B, P, H = 20, 4096, 2, 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
new = lambda *s: torch.randn(*s, device=dev)
x, wih, bias = new(T, B, P), 0.1 * new(4 * H, P), 0.1 * new(4 * H)
whht = 0.08 * new(H, 4 * H)
gates, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
gx = 0.1 * new(T, B, 4 * H)
st = torch.cuda.current_stream().cuda_stream
vp = C.c_void_p


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


for path in sorted(glob.glob(os.path.join(ROOT, "tools", "exp", "liblstm_*.so"))):
    lib = C.CDLL(path)
    f = lib.kvae_lstm128_forward
    f.argtypes = [C.c_int] * 3 + [vp] * 9 + [C.c_int, vp]
    f.restype = C.c_int
    t2 = timed(lambda: f(T, B, P, x.data_ptr(), wih.data_ptr(), bias.data_ptr(), whht.data_ptr(),
                         gates.data_ptr(), h.data_ptr(), c.data_ptr(), None, None, 1, st))
    g2 = gx.clone()
    t0 = timed(lambda: f(T, B, 0, None, None, None, whht.data_ptr(), g2.data_ptr(), h.data_ptr(),
                         c.data_ptr(), None, None, 1, st))
    print(f"{os.path.basename(path):28s} P=2 {t2:7.1f} us   P=0 {t0:7.1f} us", flush=True)
