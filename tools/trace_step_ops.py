"""Which autograd/ATen operations launch the small glue kernels of one training step.

Runs the headline step eagerly under torch.profiler and prints, for every device kernel
that is not one of this library's, the operator that launched it, its input shapes and the
Python frames above it.  Usage: python tools/trace_step_ops.py [B] [net]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from kalman_vae_amd import SampleControl, kalman_elbo

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
net = sys.argv[2] if len(sys.argv) > 2 else "lstm"
dev = torch.device("cuda:0")
torch.manual_seed(0)
ssm = bench.build_model(dev, torch.float32, net)
a = bench.bouncing_ball_observations(20, B, 2, dev, seed=1000).float().requires_grad_(True)
sc = SampleControl()
minus_one = torch.full((), -1.0, device=dev)


def step():
    ssm.zero_grad(set_to_none=True)
    a.grad = None
    obj, _ = kalman_elbo(ssm, a, sc, burn_in=3)
    obj.backward(minus_one)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
             with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

events = prof.events()
by_corr = {}
for e in events:
    if e.device_type == torch.autograd.DeviceType.CPU:
        for k in e.kernels:
            by_corr.setdefault(id(k), (k, e))
rows = []
for k, e in by_corr.values():
    if "kvae::" in k.name or k.name.startswith("Cijk"):
        continue
    stack = [s for s in (e.stack or []) if "kalman_vae_amd" in s or "bench" in s][:3]
    rows.append((e.time_range.start, k.name[:60], e.name, str(e.input_shapes)[:80], k.duration, stack))
rows.sort()
for _, kn, op, shp, dur, stack in rows:
    print(f"{dur:7.1f} us  {op:28s} {shp:80s} {kn}")
    for s in stack:
        print("             ", s)
