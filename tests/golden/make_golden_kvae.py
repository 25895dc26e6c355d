#!/usr/bin/env python3
"""Golden fixtures for the full model (``KalmanVariationalAutoencoder.elbo`` called
with frames ``xs``) made by running the reference itself (build container only):

    python tests/golden/make_golden_kvae.py [--reference /root/reference]

Stored per case: the model's state_dict, the frames, every standard-normal
draw in the order the reference consumed them (encoder sample, filter, smoother,
ELBO sample), the objective, every scalar term of the returned info and the
gradient of the objective w.r.t. every parameter.  Only data is written.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from golden_util import seeded_tensor  # noqa: E402

CASES = {
    "gaussian_lstm_f64": dict(decoder="gaussian", net="lstm", seed=51, T=5, B=3, image=(16, 16),
                              channels=1, p=2, n=4, K=3, state_transition="sample",
                              observation="sample", kl_weight=0.0, burn_in=0, train=True,
                              encoder="sample",
                              sequence_operation="mean", batch_operation="mean"),
    "bernoulli_mlp_f64": dict(decoder="bernoulli", net="mlp", seed=52, T=6, B=2, image=(16, 8),
                              channels=2, p=2, n=3, K=2, state_transition="mean",
                              observation="mean", kl_weight=0.25, burn_in=2, train=False,
                              encoder="mean",
                              sequence_operation="sum", batch_operation="mean",
                              reconst_weight=0.5, regularization_weight=0.7, kalman_weight=1.3),
    # the frame size of examples/train.py (64x64 binary frames, Bernoulli decoder, LSTM weight
    # network) with a small batch.  To keep the file small: the float64 run starts from
    # float32-representable parameters, everything is stored in float32, the LSTM matrices are
    # regenerated from a seed instead of stored, and of their gradients every 8th element
    # and the sum are kept.
    "c5_bernoulli_lstm_64x64_f64s": dict(decoder="bernoulli", net="lstm", seed=53, T=6, B=2,
                                         image=(64, 64), channels=1, p=2, n=4, K=3,
                                         state_transition="sample", observation="sample",
                                         kl_weight=0.1, burn_in=1, train=True, encoder="sample",
                                         sequence_operation="mean", batch_operation="mean",
                                         binary_frames=True, store_f32=True, seeded_lstm=True,
                                         grad_stride=8),
}
SCALARS = ["reconstruction", "regularization", "kl", "kalman_observation_log_likelihood",
           "kalman_state_transition_log_likelihood", "kalman_posterior_log_likelihood"]


def build(cfg):
    import torch.distributions.multivariate_normal as mvn_mod
    import torch.distributions.normal as normal_mod
    from kvae.kalman_vae import KalmanVariationalAutoencoder
    from kvae.sample_control import SampleControl

    torch.manual_seed(cfg["seed"])
    model = KalmanVariationalAutoencoder(
        image_size=cfg["image"], image_channels=cfg["channels"], a_dim=cfg["p"], z_dim=cfg["n"],
        K=cfg["K"], init_transition_reg_weight=0.9, init_observation_reg_weight=0.9,
        learn_noise_covariance=True, init_noise_scale=0.7,
        dynamics_parameter_network=cfg["net"], decoder_type=cfg["decoder"]).double()
    if cfg.get("store_f32", False):
        with torch.no_grad():
            for k, v in model.named_parameters():
                if cfg.get("seeded_lstm", False) and ".lstm." in k:
                    v.copy_(seeded_tensor(k, v.shape, cfg["seed"], 128 ** -0.5).double())
                else:
                    v.copy_(v.float().double())
    model.train(cfg["train"])
    T, B = cfg["T"], cfg["B"]
    xs = torch.rand(T, B, cfg["channels"], *cfg["image"], dtype=torch.float64)
    if cfg["decoder"] == "bernoulli":
        xs = (xs > 0.5).double()
    sc = SampleControl(encoder=cfg["encoder"], decoder="mean", state_transition=cfg["state_transition"],
                       observation=cfg["observation"])
    draws = []
    orig = mvn_mod._standard_normal

    def rec(shape, dtype, device):
        e = orig(shape, dtype=dtype, device=device)
        draws.append(e)
        return e

    mvn_mod._standard_normal = rec
    normal_mod._standard_normal = rec
    try:
        objective, info = model.elbo(
            sample_control=sc, xs=xs, kl_weight=cfg["kl_weight"], burn_in=cfg["burn_in"],
            reconst_weight=cfg.get("reconst_weight", 0.3),
            regularization_weight=cfg.get("regularization_weight", 1.0),
            kalman_weight=cfg.get("kalman_weight", 1.0),
            sequence_operation=cfg["sequence_operation"], batch_operation=cfg["batch_operation"])
    finally:
        mvn_mod._standard_normal = orig
        normal_mod._standard_normal = orig
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(objective, [p for _, p in model.named_parameters() if p.requires_grad],
                                allow_unused=True)
    data = {"config": np.array(json.dumps(cfg)), "in.xs": xs.numpy(),
            "out.objective": objective.detach().numpy(), "out.as": info["as"].detach().numpy(),
            "out.means": info["means"].detach().numpy(), "n_draws": np.array(len(draws))}
    for i, e in enumerate(draws):
        data[f"eps.{i}"] = e.numpy()
    for k in SCALARS:
        data["out." + k] = torch.as_tensor(info[k]).detach().double().numpy()
    for k, v in model.state_dict().items():
        if not (cfg.get("seeded_lstm", False) and ".lstm." in k):
            data["sd." + k] = v.detach().numpy()
    stride = cfg.get("grad_stride", 1)
    for k, g in zip(names, grads):
        if g is None:
            continue
        if stride > 1 and ".lstm." in k:
            data["gradsum." + k] = g.sum().numpy()
            data["gradsub." + k] = g.reshape(-1)[::stride].numpy()
        else:
            data["grad." + k] = g.numpy()
    if cfg.get("binary_frames", False):
        data["in.xs"] = data["in.xs"].astype(np.uint8)
    if cfg.get("store_f32", False):
        data = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64
                    and k != "config" else v) for k, v in data.items()}
    return data


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    os.makedirs(os.path.join(HERE, "kvae"), exist_ok=True)
    for name, cfg in CASES.items():
        if args.only and args.only not in name:
            continue
        np.savez_compressed(os.path.join(HERE, "kvae", name + ".npz"), **build(cfg))
        print("wrote", f"kvae/{name}.npz")
