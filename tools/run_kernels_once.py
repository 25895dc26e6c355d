"""Launch each of the six HIP kernels a few times at a given shape (used under
rocprofv3 to collect per-kernel counters).  usage: run_kernels_once.py B [T n p K iters]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = [int(x) for x in sys.argv[1:]]
B = a[0] if a else 4096
T, n, p, K, iters = (a[1:] + [20, 4, 2, 3, 3][len(a) - 1:]) if len(a) > 1 else (20, 4, 2, 3, 3)
dev = torch.device("cuda:0")
ssm = bench.build_model(dev, torch.float32, "mlp", n, p, K)
x = bench.bouncing_ball_observations(T, B, p, dev, seed=1)
kt = bench.time_kernels(ssm, x, 3, iters=iters, synthetic_alpha=True)
print({k: round(v * 1e6, 1) for k, v in kt.items()}, "us per launch")
