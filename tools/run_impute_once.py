#!/usr/bin/env python3
"""A few calls of the one-launch masked filter + smoother at the c3 chunk size (for rocprofv3
counter runs).  usage: run_impute_once.py [sequences]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kalman_vae_amd import SampleControl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
torch.manual_seed(0)
ssm = bench.build_model(dev, torch.float32, "mlp", 4, 2, 3).eval()
a = bench.bouncing_ball_observations(20, B, 2, dev, seed=2000).float()
mask = (torch.rand(20, B, device=dev) < 0.7).float()
mask[0] = 1.0
sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")
with torch.no_grad():
    for _ in range(4):
        ssm.impute(a, sc, observation_mask=mask)
torch.cuda.synchronize()
