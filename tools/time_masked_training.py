#!/usr/bin/env python3
"""Masked filter under autograd, forward + backward, HIP events: with the perceptron weight
network the single-node path (one launch per direction) against the step-by-step path (T
one-step launches plus torch glue); with the default LSTM network the single-node path (the
steps walked by hand through the library's kernels) against the step-by-step path (library
LSTM call and autograd nodes per step).  python tools/time_masked_training.py [B ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kalman_vae_amd import SampleControl, StateSpaceModel  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    T = 20
    for B in [int(x) for x in sys.argv[1:]] or [4096, 65536]:
        torch.manual_seed(0)
        ssm = StateSpaceModel(2, 4, 3, "mlp").to(dev).train()
        a = torch.randn(T, B, 2, device=dev, requires_grad=True)
        mask = (torch.rand(T, B, device=dev) < 0.7).float()
        sc = SampleControl()

        def step():
            ssm.zero_grad(set_to_none=True)
            a.grad = None
            out = ssm.kalman_filter(a, sc, observation_mask=mask)
            (out[0].sum() + out[1].sum() + out[6].sum()).backward()

        ssm.in_kernel_weight_network = True
        fast = timed(step)
        ssm.in_kernel_weight_network = False
        slow = timed(step, iters=3)
        print(f"B={B}: perceptron: one launch per direction {fast:.3f} ms, step-by-step {slow:.3f} ms, "
              f"ratio {slow / fast:.1f}x", flush=True)
        torch.manual_seed(0)
        ssm = StateSpaceModel(2, 4, 3, "lstm").to(dev).train()
        fast = timed(step)
        ssm.weight_model.use_hip_recurrence = False
        slow = timed(step, iters=3)
        print(f"B={B}: LSTM: one node {fast:.3f} ms, step-by-step {slow:.3f} ms, "
              f"ratio {slow / fast:.1f}x", flush=True)


if __name__ == "__main__":
    main()
