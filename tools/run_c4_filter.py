"""Run the six kernels once at config-4 shapes with a short T (for counter collection)."""
import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
T, B = int(sys.argv[1]) if len(sys.argv) > 1 else 50, 1024
ssm = bench.build_model(dev, torch.float32, "mlp", 32, 16, 8)
a = bench.bouncing_ball_observations(T, B, 16, dev, seed=1)
kt = bench.time_kernels(ssm, a, 3, iters=1, synthetic_alpha=True)
print({k: round(v * 1e6 / T, 2) for k, v in kt.items()}, "us per step")
