"""One masked training step (filter forward + backward, LSTM network, B sequences) a few times,
for rocprofv3 --kernel-trace --stats; prints the wall time per step by HIP events too.
python tools/exp/profile_masked_lstm.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kalman_vae_amd import SampleControl, StateSpaceModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
ssm = StateSpaceModel(2, 4, 3, "lstm").to(dev).train()
a = torch.randn(20, B, 2, device=dev, requires_grad=True)
mask = (torch.rand(20, B, device=dev) < 0.7).float()
sc = SampleControl()


def step():
    ssm.zero_grad(set_to_none=True)
    a.grad = None
    out = ssm.kalman_filter(a, sc, observation_mask=mask)
    (out[0].sum() + out[1].sum() + out[6].sum()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    step()
e1.record()
torch.cuda.synchronize()
print(f"B={B}: {e0.elapsed_time(e1) / 10:.3f} ms per step (13 steps run in all)")
