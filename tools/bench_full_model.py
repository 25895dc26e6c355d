#!/usr/bin/env python3
"""Full Kalman-VAE training step (BASELINE config 5): conv encoder on 64x64 frames ->
Kalman filter / smoother / ELBO (HIP kernels) -> conv decoder, backward, one flat
gradient all-reduce, Adam.  One process per GPU:

    python tools/bench_full_model.py [--batch 256] [--steps 20] [--warmup 5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/bench_full_model.py ...

Prints one JSON line (frames/s and sequence-steps/s over all ranks, max time over ranks).
The encoder/decoder are stock PyTorch-ROCm modules, as in the reference; this measures
how much of a real training step the Kalman path is, not a headline number.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalman_vae_amd import KalmanVariationalAutoencoder, SampleControl  # noqa: E402
from kalman_vae_amd.parallel import DataParallelStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--image", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--net", default="lstm", choices=["lstm", "mlp"])
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(1234)          # identical initial parameters on every rank
    model = KalmanVariationalAutoencoder(
        image_size=(args.image, args.image), image_channels=1, a_dim=2, z_dim=4, K=3,
        init_transition_reg_weight=0.9, init_observation_reg_weight=0.9,
        learn_noise_covariance=False, init_noise_scale=1.0,
        dynamics_parameter_network=args.net, decoder_type="bernoulli").to(dev).train()
    step = DataParallelStep(model, torch.optim.Adam(model.parameters(), lr=1e-3))
    torch.manual_seed(99 + rank)     # every rank its own synthetic frames
    xs = (torch.rand(args.seq_len, args.batch, 1, args.image, args.image, device=dev) > 0.8).float()
    sc = SampleControl()

    def objective():
        return model.elbo(sc, xs=xs, burn_in=3)[0]

    for _ in range(args.warmup):
        step(objective)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step(objective)
    torch.cuda.synchronize()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        s = float(elapsed) / args.steps
        print(json.dumps({
            "workload": f"full KVAE training step, {args.image}x{args.image} frames, T={args.seq_len}, "
                        f"{args.batch} sequences per GPU, {args.net} weight network, Adam, fp32",
            "n_gpus": world, "ms_per_step": 1e3 * s,
            "frames_per_s": args.seq_len * args.batch * world / s,
            "sequence_steps_per_s": args.seq_len * args.batch * world / s,
            "objective": float(last), "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
