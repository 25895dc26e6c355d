#!/usr/bin/env python3
"""Launch each kernel of the LSTM weight network a few times (register-resident layer kernels,
the row-streaming products, the weight-gradient product) for rocprofv3 counter runs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kalman_vae_amd import _capi as capi   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T, P, H = 20, 2, 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
lib = capi.load()
new = lambda *s: torch.randn(*s, device=dev)
x, wih, bias = new(T, B, P), 0.1 * new(4 * H, P), 0.1 * new(4 * H)
whht, whh = 0.08 * new(H, 4 * H), 0.08 * new(4 * H, H)
gates, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
gx = new(T, B, 4 * H)
g_h = 1e-3 * new(T, B, H)
dG, dx = new(T, B, 4 * H), new(T, B, P)
part = new(lib.kvae_lstm128_blocks(B), 4 * H * (2 + P))
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.kvae_lstm128_forward(T, B, P, x.data_ptr(), wih.data_ptr(), bias.data_ptr(),
                             whht.data_ptr(), gates.data_ptr(), h.data_ptr(), c.data_ptr(), None,
                             None, 1, st)
    gx2 = gx.clone()
    lib.kvae_lstm128_forward(T, B, 0, None, None, None, whht.data_ptr(), gx2.data_ptr(),
                             h.data_ptr(), c.data_ptr(), None, None, 1, st)
    lib.kvae_lstm128_backward(T, B, P, x.data_ptr(), wih.data_ptr(), gates.data_ptr(), c.data_ptr(),
                              None, g_h.data_ptr(), whh.data_ptr(), dG.data_ptr(), dx.data_ptr(),
                              part.data_ptr(), st)
    lib.kvae_lstm128_backward(T, B, 0, None, None, gates.data_ptr(), c.data_ptr(), None,
                              g_h.data_ptr(), whh.data_ptr(), dG.data_ptr(), None, part.data_ptr(),
                              st)
    KH = 3
    al = torch.softmax(new(T, B, KH), -1)
    gal = 1e-3 * new(T, B, KH)
    wh = 0.1 * new(KH, H)
    hpart = new(lib.kvae_lstm128_blocks(B), KH * (H + 1))
    lib.kvae_lstm128_backward_head(T, B, KH, gates.data_ptr(), c.data_ptr(), None, al.data_ptr(),
                                   gal.data_ptr(), wh.data_ptr(), whh.data_ptr(), dG.data_ptr(),
                                   part.data_ptr(), hpart.data_ptr(), st)
    R = T * B
    wih1 = 0.08 * new(4 * H, H)
    wih1t = wih1.t().contiguous()
    gh0 = new(R, H)
    S = lib.kvae_lstm128_wgrad_slabs(R)
    slabs = new(S, 4 * H * H)
    lib.kvae_rowstream_matmul(R, H, 4 * H, h.data_ptr(), wih1t.data_ptr(), bias.data_ptr(),
                              gx2.data_ptr(), st)
    lib.kvae_rowstream_matmul(R, 4 * H, H, dG.data_ptr(), wih1.data_ptr(), None, gh0.data_ptr(), st)
    lib.kvae_lstm128_wgrad(R, dG.data_ptr(), h.data_ptr(), B, slabs.data_ptr(), 4 * H * H, st)
torch.cuda.synchronize()
