#!/usr/bin/env python3
"""Generate golden fixtures by running the reference implementation itself.

Run in the build container only (the reference checkout does not exist on the
GPU box):

    python tests/golden/make_golden.py [--reference /root/reference]

The script imports the reference package, runs
``KalmanVariationalAutoencoder.elbo(as_=...)`` (which calls the reference's
``kalman_filter`` and ``kalman_smooth``) on small seeded inputs, and stores
inputs, the standard-normal draws in draw order, every output and the
gradients of a fixed scalar loss as ``tests/golden/*.npz``.  Only data is
written; no reference source is copied.

Loss whose gradients are stored:
    loss = objective + sum_i <w_i, output_i>
with w_i seeded random cotangents for the four filter mean/cov outputs, the
smoother's means/covariances/zs/as_resampled, mat_As, mat_Cs and
as_for_weight; the cotangents are stored too.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _np(t):
    return t.detach().cpu().numpy()


def build_case(ref, name, cfg):
    import torch.distributions.multivariate_normal as mvn_mod
    from kvae import kalman_vae as kvae_mod
    from kvae.kalman_vae import KalmanVariationalAutoencoder
    from kvae.sample_control import SampleControl
    from kvae.state_space_model import StateSpaceModel

    dtype = torch.float64 if cfg["dtype"] == "float64" else torch.float32
    torch.manual_seed(cfg["seed"])
    T, B, p, n, K = cfg["T"], cfg["B"], cfg["p"], cfg["n"], cfg["K"]

    m0 = P0 = None
    if cfg.get("general_init", False):
        m0 = 0.3 * torch.randn(n)
        W = torch.randn(n, n)
        P0 = 0.5 * torch.eye(n) + 0.1 * W @ W.T
        if cfg.get("symmetric_init", False):
            # W @ W.T in float32 is symmetric only up to rounding (1e-8); the reference inverts
            # the full innovation matrix, a Cholesky-based inverse reads one triangle
            P0 = 0.5 * (P0 + P0.T)
    ssm = StateSpaceModel(
        a_dim=p, z_dim=n, K=K,
        dynamics_parameter_network=cfg["net"],
        hidden_dim=cfg.get("hidden_dim", 8),
        num_layers=cfg.get("num_layers", 2),
        initial_state_mean=m0, initial_state_covariance=P0,
        fix_matrices=cfg.get("fix_matrices", False),
        learn_noise_covariance=cfg.get("learn_noise_covariance", False),
        init_transition_reg_weight=cfg.get("init_reg", 0.9),
        init_observation_reg_weight=cfg.get("init_reg", 0.9),
    )
    if cfg.get("seeded_lstm", False):
        # the reference's default network (2 x 128) has 200 k parameters: its large matrices
        # are regenerated from a seed on both sides instead of being stored, and of their
        # gradients every 16th element and the sum are kept
        sys.path.insert(0, os.path.dirname(HERE))
        from golden_util import seeded_tensor
        seeded = {}
        with torch.no_grad():
            for k, v in ssm.weight_model.named_parameters():
                if v.numel() >= 4096:
                    bound = 1.0 / float(cfg["hidden_dim"]) ** 0.5
                    v.copy_(seeded_tensor("weight_model." + k, v.shape, cfg["seed"], bound))
                    seeded["weight_model." + k] = [list(v.shape), bound]
        cfg = dict(cfg, seeded=seeded)
    if cfg.get("general_noise", False):
        with torch.no_grad():
            ssm._mat_Q_L.copy_(0.6 * torch.eye(n) + 0.15 * torch.randn(n, n))
            ssm._mat_R_L.copy_(0.5 * torch.eye(p) + 0.15 * torch.randn(p, p))
    if cfg.get("identity_noise_via_setter", False):
        ssm.mat_Q = torch.eye(n)
        ssm.mat_R = torch.eye(p)
    model = KalmanVariationalAutoencoder(
        image_size=(8, 8), image_channels=1, a_dim=p, z_dim=n, K=K,
        init_transition_reg_weight=0.9, init_observation_reg_weight=0.9,
        learn_noise_covariance=False, init_noise_scale=1.0,
        dynamics_parameter_network="mlp")
    model.state_space_model = ssm
    model = model.to(dtype)
    ssm = model.state_space_model
    if cfg["train_mode"]:
        model.train()
    else:
        model.eval()

    a = (0.7 * torch.randn(T, B, p)).to(dtype).requires_grad_(True)
    mask = None
    if cfg["mask"] == "random":
        mask = (torch.rand(T, B) > 0.3).to(dtype)
        mask[0] = 1.0
        mask[-1] = 1.0
    elif cfg["mask"] == "continuous":
        mask = torch.ones(T, B, dtype=dtype)
        mask[2:2 + max(1, T // 3)] = 0.0
    elif cfg["mask"] == "fractional":
        mask = torch.rand(T, B).to(dtype)
    sc = SampleControl(encoder="sample", decoder="mean",
                       state_transition=cfg["state_transition"],
                       observation=cfg["observation"])

    # record standard-normal draws and the un-aggregated ELBO terms
    draws = []
    orig_sn = mvn_mod._standard_normal

    def rec_sn(shape, dtype, device):
        e = orig_sn(shape, dtype=dtype, device=device)
        draws.append(e)
        return e

    terms = []
    orig_agg = kvae_mod.aggregate

    def rec_agg(value, **kw):
        terms.append(value)
        return orig_agg(value, **kw)

    mvn_mod._standard_normal = rec_sn
    kvae_mod.aggregate = rec_agg
    try:
        objective, info = model.elbo(
            sample_control=sc, as_=a, observation_mask=mask,
            learn_weight_model=cfg.get("learn_weight_model", True),
            symmetrize_covariance=cfg.get("symmetrize", True),
            burn_in=cfg.get("burn_in", 0),
            sequence_operation=cfg.get("sequence_operation", "mean"),
            batch_operation=cfg.get("batch_operation", "mean"),
            kalman_weight=cfg.get("kalman_weight", 1.0))
    finally:
        mvn_mod._standard_normal = orig_sn
        kvae_mod.aggregate = orig_agg

    # decode the draw order: filter (per t: state?, obs?), smoother (t=T-1..0:
    # state?, obs?), ELBO (T,B,n)
    st, ob = sc.state_transition == "sample", sc.observation == "sample"
    it = iter(draws)
    z = lambda *s: torch.zeros(*s, dtype=dtype)
    f_eps_s, f_eps_o = z(T, B, n), z(T, B, p)
    s_eps_s, s_eps_o = z(T, B, n), z(T, B, p)
    for t in range(T):
        if st:
            f_eps_s[t] = next(it)
        if ob:
            f_eps_o[t] = next(it)
    for t in reversed(range(T)):
        if st:
            s_eps_s[t] = next(it)
        if ob:
            s_eps_o[t] = next(it)
    e_eps = next(it)
    assert next(it, None) is None, "unexpected extra random draws"
    assert e_eps.shape == (T, B, n)

    out_names = ["filter_means", "filter_covariances", "filter_next_means",
                 "filter_next_covariances", "mat_As", "mat_Cs", "filter_as",
                 "means", "covariances", "zs", "as_resampled"]
    g = torch.Generator().manual_seed(1000 + cfg["seed"])
    cots = {k: torch.randn(info[k].shape, generator=g).to(dtype) for k in out_names}
    loss = objective
    for k in out_names:
        # "objective_only": the gradients a training step takes (examples/train.py:128-131
        # of the reference differentiates -elbo and nothing else)
        if info[k].requires_grad and not cfg.get("objective_only", False):
            loss = loss + (cots[k] * info[k]).sum()
    params = {"as_": a, "_mat_A_K": ssm._mat_A_K, "_mat_C_K": ssm._mat_C_K,
              "_mat_Q_L": ssm._mat_Q_L, "_mat_R_L": ssm._mat_R_L}
    for k, v in ssm.weight_model.named_parameters():
        params["weight_model." + k] = v
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)

    data = {"config": np.array(json.dumps(cfg))}
    data["in.as_"] = _np(a)
    if mask is not None:
        data["in.mask"] = _np(mask)
    data["in.Q"] = _np(ssm.mat_Q)
    data["in.R"] = _np(ssm.mat_R)
    data["in.Q_reg"] = np.array(ssm.Q_reg)
    data["in.R_reg"] = np.array(ssm.R_reg)
    data["in.m0"] = _np(ssm.initial_state_mean)
    data["in.P0"] = _np(ssm.initial_state_covariance)
    for k, v in ssm.state_dict().items():
        if k not in cfg.get("seeded", {}):
            data["sd." + k] = _np(v)
    data["eps.filter_state"] = _np(f_eps_s)
    data["eps.filter_obs"] = _np(f_eps_o)
    data["eps.smooth_state"] = _np(s_eps_s)
    data["eps.smooth_obs"] = _np(s_eps_o)
    data["eps.elbo"] = _np(e_eps)
    for k in out_names + ["weights"]:
        data["out." + k] = _np(info[k])
    data["out.elbo_obs"] = _np(terms[0])
    data["out.elbo_trans"] = _np(terms[1])
    data["out.elbo_post"] = _np(terms[2])
    data["out.agg_obs"] = _np(info["kalman_observation_log_likelihood"])
    data["out.agg_trans"] = _np(info["kalman_state_transition_log_likelihood"])
    data["out.agg_post"] = _np(info["kalman_posterior_log_likelihood"])
    data["out.objective"] = _np(objective)
    if not cfg.get("objective_only", False):
        for k in out_names:
            data["cot." + k] = _np(cots[k])
    for (k, _), gval in zip(params.items(), grads):
        if gval is None:
            continue
        if k in cfg.get("seeded", {}):
            data["gradsub." + k] = _np(gval.reshape(-1)[::16])
            data["gradsum." + k] = _np(gval.sum())
        else:
            data["grad." + k] = _np(gval)
    if cfg.get("store_f32", False):
        # large-state fixtures: computed in float64, stored in float32 (inputs are
        # exactly representable; outputs only need ~1e-7 for the fp32 1e-4 bar)
        data = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64
                    and k != "config" else v) for k, v in data.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **data)
    return path


def cases():
    base = dict(T=7, B=3, p=2, n=4, K=3, net="mlp", dtype="float64", seed=0,
                train_mode=False, mask="none", state_transition="mean",
                observation="mean")
    c = {}
    # G1 deterministic core, MLP and LSTM weight nets, both precisions
    c["g1_mlp_f64"] = dict(base)
    c["g1_lstm_f64"] = dict(base, net="lstm", seed=1)
    c["g1_lstm_f32"] = dict(base, net="lstm", seed=1, dtype="float32")
    # G2 masks (eval, mean modes)
    c["g2_random_mask_f64"] = dict(base, net="lstm", mask="random", seed=2, T=9, B=4)
    c["g2_continuous_mask_f64"] = dict(base, net="mlp", mask="continuous", seed=3, T=9)
    c["g2_random_mask_f32"] = dict(base, net="lstm", mask="random", seed=2, T=9, B=4,
                                   dtype="float32")
    # G3 train mode, all sampling paths
    c["g3_sample_f64"] = dict(base, net="lstm", train_mode=True, seed=4,
                              state_transition="sample", observation="sample")
    c["g3_sample_mask_f64"] = dict(base, net="mlp", train_mode=True, seed=5, mask="random",
                                   state_transition="sample", observation="sample")
    # G4 gradient gates
    c["g4_burnin3_f64"] = dict(base, net="lstm", seed=6, burn_in=3, train_mode=True,
                               state_transition="sample", observation="sample")
    c["g4_nosym_f64"] = dict(base, seed=7, symmetrize=False)
    c["g4_nolearnw_f64"] = dict(base, net="lstm", seed=8, learn_weight_model=False, burn_in=2)
    c["g4_learnnoise_f64"] = dict(base, seed=9, learn_noise_covariance=True,
                                  general_noise=True, burn_in=1, mask="random")
    c["g4_fixmat_f64"] = dict(base, seed=10, fix_matrices=True)
    # G5 aggregation variants
    c["g5_sum_sum_f64"] = dict(base, seed=11, sequence_operation="sum", batch_operation="sum",
                               kalman_weight=0.7)
    c["g5_mean_sum_f64"] = dict(base, seed=12, sequence_operation="mean", batch_operation="sum")
    c["g5_sum_mean_f64"] = dict(base, seed=13, sequence_operation="sum", batch_operation="mean")
    # G6 fractional mask
    c["g6_fractional_mask_f64"] = dict(base, net="mlp", seed=14, mask="fractional",
                                       learn_noise_covariance=True, general_noise=True)
    # G8 general parameters
    c["g8_general_f64"] = dict(base, net="lstm", seed=15, general_noise=True, general_init=True,
                               learn_noise_covariance=True, train_mode=True,
                               state_transition="sample", observation="sample", init_reg=0.6)
    c["g8_general_f32"] = dict(c["g8_general_f64"], dtype="float32")
    c["g8_setter_identity_f64"] = dict(base, seed=16, identity_noise_via_setter=True, burn_in=3,
                                       T=20, B=2, net="lstm", train_mode=True,
                                       state_transition="sample", observation="sample")
    # other small dimensions exercised by the templated kernels
    c["dims_n2p1k2_f64"] = dict(base, seed=17, n=2, p=1, K=2, mask="random")
    c["dims_n3p2k2_f64"] = dict(base, seed=18, n=3, p=2, K=2, general_noise=True,
                                learn_noise_covariance=True)
    c["dims_n8p4k4_f64"] = dict(base, seed=19, n=8, p=4, K=4, T=5, B=2, net="lstm",
                                general_noise=True, learn_noise_covariance=True,
                                train_mode=True, state_transition="sample",
                                observation="sample")
    # G7 large state (wavefront-per-sequence kernels): float64 run stored as float32
    c["g7_large_f64s"] = dict(base, seed=20, n=32, p=16, K=8, T=6, B=2, net="lstm",
                              mask="random", general_noise=True, learn_noise_covariance=True,
                              train_mode=True, state_transition="sample", observation="sample",
                              burn_in=1, store_f32=True)
    c["g7_mid_f64"] = dict(base, seed=21, n=16, p=8, K=4, T=6, B=2, net="mlp",
                           general_noise=True, learn_noise_covariance=True, train_mode=True,
                           state_transition="sample", observation="sample", burn_in=1)
    c["dims_n6p3k2_f64"] = dict(base, seed=22, n=6, p=3, K=2, mask="random",
                                general_noise=True, learn_noise_covariance=True)
    # the training call (examples/train.py:115-131): train mode, every path sampled, no mask,
    # burn-in, gradients of the objective alone
    tr = dict(base, train_mode=True, state_transition="sample", observation="sample",
              objective_only=True, burn_in=3, T=20, B=4, identity_noise_via_setter=True)
    c["train_lstm_f64"] = dict(tr, net="lstm", seed=30)
    c["train_lstm_f32"] = dict(tr, net="lstm", seed=30, dtype="float32")
    # the reference's default weight network (hidden_dim 128, two layers)
    c["train_lstm128_f64"] = dict(tr, net="lstm", seed=35, hidden_dim=128, seeded_lstm=True, B=6)
    c["train_lstm128_f32"] = dict(tr, net="lstm", seed=35, hidden_dim=128, seeded_lstm=True, B=6,
                                  dtype="float32")
    # masked sequences with that network under autograd (the imputation feeds the LSTM step by
    # step): every output and every gradient, draws on both paths, burn-in, learned noise
    c["masked_lstm128_f64"] = dict(base, net="lstm", seed=36, hidden_dim=128, seeded_lstm=True,
                                   T=9, B=5, mask="random", train_mode=True,
                                   state_transition="sample", observation="sample", burn_in=1,
                                   general_noise=True, learn_noise_covariance=True)
    c["masked_lstm128_mean_f64"] = dict(base, net="lstm", seed=37, hidden_dim=128,
                                        seeded_lstm=True, T=8, B=4, mask="random")
    c["train_mlp_warmup_f64"] = dict(tr, net="mlp", seed=31, learn_weight_model=False, T=9)
    c["train_general_f64"] = dict(base, net="lstm", seed=32, train_mode=True,
                                  state_transition="sample", observation="sample",
                                  objective_only=True, burn_in=2, T=9, B=5, general_noise=True,
                                  general_init=True, symmetric_init=True,
                                  learn_noise_covariance=True, init_reg=0.8,
                                  sequence_operation="sum", kalman_weight=0.8)
    c["train_nosym_n3_f64"] = dict(base, seed=33, n=3, p=2, K=2, train_mode=True,
                                   state_transition="sample", observation="sample",
                                   objective_only=True, symmetrize=False, T=8, B=3,
                                   general_noise=True, learn_noise_covariance=True)
    c["train_n8_f64"] = dict(base, seed=34, n=8, p=4, K=4, T=6, B=2, net="lstm", train_mode=True,
                             state_transition="sample", observation="sample",
                             objective_only=True, burn_in=1)
    # (z_dim, a_dim, K) without a kernel of their own: solved as a block of a larger compiled
    # problem (the reference constructor, state_space_model.py:16-33, takes any dimensions)
    c["embed_n5p3k4_f64"] = dict(base, seed=40, n=5, p=3, K=4, T=6, B=3, net="lstm",
                                 mask="random", general_noise=True, learn_noise_covariance=True,
                                 general_init=True, symmetric_init=True, train_mode=True,
                                 state_transition="sample", observation="sample", burn_in=1)
    c["embed_n4p2k5_f64"] = dict(base, seed=41, K=5, general_noise=True,
                                 learn_noise_covariance=True)
    c["embed_n4p2k1_f64"] = dict(base, seed=42, K=1, net="lstm", train_mode=True,
                                 state_transition="sample", observation="sample", burn_in=2)
    c["embed_n10p5k3_f64"] = dict(base, seed=43, n=10, p=5, K=3, T=5, B=2, net="lstm",
                                  general_noise=True, learn_noise_covariance=True,
                                  train_mode=True, state_transition="sample",
                                  observation="sample")
    c["embed_n2p3k2_mask_f64"] = dict(base, seed=44, n=2, p=3, K=2, mask="continuous", T=9)
    c["train_embed_n4p2k6_f64"] = dict(tr, net="lstm", seed=45, K=6, T=10, B=3)
    c["train_embed_n5p3k2_f64"] = dict(base, seed=46, n=5, p=3, K=2, train_mode=True,
                                       state_transition="sample", observation="sample",
                                       objective_only=True, burn_in=1, T=8, B=3,
                                       general_noise=True, general_init=True,
                                       symmetric_init=True, learn_noise_covariance=True,
                                       sequence_operation="sum", kalman_weight=0.6)
    # G7 at T=64: the large-state kernels over a longer recursion (float64 run stored as
    # float32; the gradients of the objective alone keep the file small)
    c["g7_large_T64_f64s"] = dict(base, seed=47, n=32, p=16, K=8, T=64, B=1, net="lstm",
                                  general_noise=True, learn_noise_covariance=True,
                                  train_mode=True, state_transition="sample",
                                  observation="sample", burn_in=2, objective_only=True,
                                  store_f32=True)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    torch.set_num_threads(1)
    total = 0
    for name, cfg in cases().items():
        if args.only and args.only not in name:
            continue
        path = build_case(args.reference, name, cfg)
        sz = os.path.getsize(path)
        total += sz
        print(f"{name}: {sz} bytes")
    print("total bytes:", total)


if __name__ == "__main__":
    main()
