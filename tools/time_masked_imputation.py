"""Throughput of masked filter + smoother imputation (forward only, eval mode), the
shape of the reference's evaluation path (kvae/evaluation.py:100-140): Bernoulli
keep-probability 0.7 mask with first and last step observed.
usage: python tools/time_masked_imputation.py [B] [net] [stepwise]
"stepwise" steps a perceptron network between one-step filter launches instead of
evaluating it inside the filter kernel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import bouncing_ball_observations, build_model  # noqa: E402
from kalman_vae_amd import SampleControl  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
net = sys.argv[2] if len(sys.argv) > 2 else "lstm"
T = 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
ssm = build_model(dev, torch.float32, net).eval()
ssm.in_kernel_weight_network = "stepwise" not in sys.argv[3:]
a = bouncing_ball_observations(T, B, 2, dev, seed=1)
mask = (torch.rand(T, B, device=dev) < 0.7).float()
mask[0] = 1.0
mask[-1] = 1.0
sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")


def run():
    with torch.no_grad():
        out = ssm.kalman_filter(a, sc, observation_mask=mask)
        sm = ssm.kalman_smooth(a, *out[:6], sample_control=sc)
    return sm


for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
iters = 5
for _ in range(iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"masked imputation T={T} B={B} net={net} in_kernel_net={ssm.in_kernel_weight_network}: {dt * 1e3:.2f} ms per call, "
      f"{T * B / dt / 1e6:.2f} M sequence-steps/s (forward only)")
