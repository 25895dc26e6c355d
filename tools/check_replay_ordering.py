"""Is work queued after a replay of the captured training step ordered after ALL of the
graph's kernels?  Clones every gradient right after the replay (no synchronisation), again
after a device synchronisation, and compares; also looks for non-finite values.
usage: python tools/check_replay_ordering.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from kalman_vae_amd import SampleControl
from kalman_vae_amd.graphed import GraphedElboStep

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ssm = bench.build_model(dev, torch.float32, "lstm")
a = bench.bouncing_ball_observations(20, 4096, 2, dev, seed=1000).float().requires_grad_(True)
step = GraphedElboStep(ssm, a, SampleControl(), burn_in=3)
named = None
bad = 0
for i in range(iters):
    step(None)
    if named is None:
        named = [(k, p) for k, p in ssm.named_parameters() if p.grad is not None]
        named.append(("as_", step.static_as))
    early = [p.grad.clone() for _, p in named]
    torch.cuda.synchronize()
    for (k, p), x in zip(named, early):
        if not torch.isfinite(p.grad).all():
            bad += 1
            print(f"iter {i}: {k} has non-finite entries")
        elif not torch.equal(x, p.grad):
            bad += 1
            print(f"iter {i}: {k} changed after the clone: max diff {float((x - p.grad).abs().max()):.3e}")
print(f"{iters} replays, {bad} problems")
sys.exit(1 if bad else 0)
