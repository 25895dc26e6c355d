#!/usr/bin/env python3
"""A T = 1000 trace of the reference at the large state size (z_dim 32, a_dim 16, K 8) in
float64, reduced to summary statistics so that the file stays small (build container only):

    python tests/golden/make_golden_long.py [--reference /root/reference]

The reference's ``KalmanVariationalAutoencoder.elbo(as_=...)`` runs one sequence of 1000
steps ("mean" modes: the only random draw is the ELBO sample, rounded to float32 before the
reference uses it so that the float32 file holds it exactly).  Stored: the inputs, the
state_dict, the three ELBO terms per step, the objective, per-step traces of the filtered
and smoothed covariances and norms of the filtered and smoothed means, and the gradient of
the objective w.r.t. the observations, the bases and the noise factors.  Only data is
written.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(T=1000, B=1, p=16, n=32, K=8, net="lstm", hidden_dim=8, num_layers=2, seed=60,
           dtype="float64", train_mode=False, state_transition="mean", observation="mean",
           sequence_operation="mean", batch_operation="mean", init_reg=0.9)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    import torch.distributions.multivariate_normal as mvn_mod
    from kvae import kalman_vae as kvae_mod
    from kvae.kalman_vae import KalmanVariationalAutoencoder
    from kvae.sample_control import SampleControl
    from kvae.state_space_model import StateSpaceModel

    c = CFG
    torch.set_num_threads(1)
    torch.manual_seed(c["seed"])
    T, B, p, n, K = c["T"], c["B"], c["p"], c["n"], c["K"]
    ssm = StateSpaceModel(a_dim=p, z_dim=n, K=K, dynamics_parameter_network=c["net"],
                          hidden_dim=c["hidden_dim"], num_layers=c["num_layers"],
                          learn_noise_covariance=True,
                          init_transition_reg_weight=c["init_reg"],
                          init_observation_reg_weight=c["init_reg"])
    with torch.no_grad():
        ssm._mat_Q_L.copy_(0.6 * torch.eye(n) + 0.05 * torch.randn(n, n))
        ssm._mat_R_L.copy_(0.5 * torch.eye(p) + 0.05 * torch.randn(p, p))
    model = KalmanVariationalAutoencoder(
        image_size=(8, 8), image_channels=1, a_dim=p, z_dim=n, K=K,
        init_transition_reg_weight=0.9, init_observation_reg_weight=0.9,
        learn_noise_covariance=False, init_noise_scale=1.0, dynamics_parameter_network="mlp")
    model.state_space_model = ssm
    model = model.double().eval()
    ssm = model.state_space_model
    a = (0.7 * torch.randn(T, B, p)).double().requires_grad_(True)
    sc = SampleControl(encoder="sample", decoder="mean", state_transition="mean",
                       observation="mean")
    draws, terms = [], []
    orig_sn, orig_agg = mvn_mod._standard_normal, kvae_mod.aggregate

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype=dtype, device=device).float().to(dtype)
        draws.append(e)
        return e

    def rec_agg(value, **kw):
        terms.append(value)
        return orig_agg(value, **kw)

    mvn_mod._standard_normal, kvae_mod.aggregate = rec_sn, rec_agg
    try:
        objective, info = model.elbo(sample_control=sc, as_=a)
    finally:
        mvn_mod._standard_normal, kvae_mod.aggregate = orig_sn, orig_agg
    assert len(draws) == 1 and draws[0].shape == (T, B, n)
    params = {"as_": a, "_mat_A_K": ssm._mat_A_K, "_mat_C_K": ssm._mat_C_K,
              "_mat_Q_L": ssm._mat_Q_L, "_mat_R_L": ssm._mat_R_L}
    grads = torch.autograd.grad(objective, list(params.values()))
    f32 = lambda t: t.detach().numpy().astype(np.float32)
    f64 = lambda t: t.detach().numpy()
    trace = lambda P: P.diagonal(dim1=-2, dim2=-1).sum(-1)
    data = {"config": np.array(json.dumps(c)), "in.as_": f32(a), "eps.elbo": f32(draws[0])}
    for k, v in ssm.state_dict().items():
        data["sd." + k] = f32(v)
    data["in.Q_reg"], data["in.R_reg"] = np.array(ssm.Q_reg), np.array(ssm.R_reg)
    data["out.elbo_obs"], data["out.elbo_trans"], data["out.elbo_post"] = map(f64, terms[:3])
    data["out.objective"] = f64(objective)
    data["sum.filter_cov_trace"] = f64(trace(info["filter_covariances"]))
    data["sum.smooth_cov_trace"] = f64(trace(info["covariances"]))
    data["sum.filter_mean_norm"] = f64(info["filter_means"].norm(dim=-1))
    data["sum.smooth_mean_norm"] = f64(info["means"].norm(dim=-1))
    data["sum.last_filter_cov"] = f64(info["filter_covariances"][-1])
    data["sum.first_smooth_cov"] = f64(info["covariances"][0])
    for (k, _), g in zip(params.items(), grads):
        data["grad." + k] = f32(g)
    os.makedirs(os.path.join(HERE, "long"), exist_ok=True)
    path = os.path.join(HERE, "long", "t1000_n32_f64.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes; objective", float(objective))


if __name__ == "__main__":
    main()
