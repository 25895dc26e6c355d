"""RCCL process group with ONE rank on one GPU next to the graph-captured step: checks that
the communicator (and its watchdog thread) and the HIP graph capture/replay of the training
step coexist, and that the flat-bucket all-reduce after a replay leaves the gradients as they
were (world size 1: the average of one rank).  usage: python tools/rehearse_nccl_one_rank.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
from kalman_vae_amd import SampleControl
from kalman_vae_amd.graphed import GraphedElboStep
from kalman_vae_amd.parallel import FlatGradBucket

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
warm = torch.ones(4, device=dev)
dist.all_reduce(warm)                      # communicator exists before the capture
torch.cuda.synchronize()
torch.manual_seed(0)
ssm = bench.build_model(dev, torch.float32, "lstm")
a = bench.bouncing_ball_observations(20, 4096, 2, dev, seed=1000).float().requires_grad_(True)
bucket = FlatGradBucket(ssm.parameters())
step = GraphedElboStep(ssm, a, SampleControl(), burn_in=3)
for i in range(5):
    obj = step(None)
    named = [(k, p) for k, p in ssm.named_parameters() if p.grad is not None]
    before = [p.grad.clone() for _, p in named]
    bucket.all_reduce(average=True)
    for (k, p), x in zip(named, before):
        assert torch.equal(x, p.grad), (k, tuple(p.shape), int(torch.isnan(x).sum()),
                                        int(torch.isnan(p.grad).sum()), float((x - p.grad).abs().max()))
# The bucket skips the collective at world size 1.  Tell it the world has two ranks (the group
# still has one, so RCCL's AVG / SUM over it is the identity): the real RCCL call on the flat
# buffer with its agreement flags, blocking and overlapped, the work handle's wait(), the
# non-blocking copy of the flags to pinned memory and their check one call later all execute.
import kalman_vae_amd.parallel as par
real_world = par.dist.get_world_size
par.dist.get_world_size = lambda group=None: 2
try:
    for overlapped in (False, True):
        for i in range(4):
            obj = step(None)
            before = [p.grad.clone() for _, p in named]
            # (a sum over "two" ranks is what one rank's flags look like to the check as half
            # of the expected count: only the average is consistent with a group of one)
            out = bucket.all_reduce(average=True, async_op=overlapped)
            if overlapped:
                assert hasattr(out, "wait")
        bucket.wait()
        bucket.check()
        for (k, p), x in zip(named, before):
            assert torch.equal(x, p.grad), k
finally:
    par.dist.get_world_size = real_world
dist.barrier()
torch.cuda.synchronize()
print("objective", float(obj), "ok: graph replay + RCCL all-reduce, 5 steps; the bucket's collective path "
      "(blocking and overlapped) with its flags on a one-rank group")
dist.destroy_process_group()
