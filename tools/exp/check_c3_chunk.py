"""c3's masked imputation on one chunk of 1 048 576 sequences against the same sequences in
chunks of 262 144 (the inputs are four copies of one chunk): every output must be the same
numbers, whatever the chunk (rows are independent in every kernel of the path)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kalman_vae_amd import SampleControl  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
ssm = bench.build_model(dev, torch.float32, "lstm", 4, 2, 3).eval()
sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")
T, C = 20, 262144
a = bench.bouncing_ball_observations(T, C, 2, dev, seed=2000).float()
mask = (torch.rand(T, C, device=dev) < 0.7).float()
mask[0] = 1.0
mask[-1] = 1.0
with torch.no_grad():
    small = ssm.impute(a, sc, observation_mask=mask)
    big = ssm.impute(a.repeat(1, 4, 1), sc, observation_mask=mask.repeat(1, 4))
worst = 0.0
for s, b in zip(small, big):
    for q in range(4):
        part = b[:, q * C:(q + 1) * C]
        worst = max(worst, float((part - s).abs().max()))
        assert torch.equal(part, s) or float((part - s).abs().max()) < 1e-6, float((part - s).abs().max())
print("one chunk of 1 048 576 == four chunks of 262 144, largest difference", worst)
