#!/usr/bin/env python3
"""Golden fixtures for ``StateSpaceModel.predict_future`` made by running the
reference itself (build container only; the reference does not exist on the GPU box):

    python tests/golden/make_golden_predict.py [--reference /root/reference]

Eval mode: the reference filters a seeded sequence (all-"mean" sample control) and rolls
``num_steps`` steps into the future (kvae/state_space_model.py:486-575) with
state_transition "mean" or "sample"; for "sample" the standard-normal draws are recorded in
draw order (``eps``, one (B, z_dim) draw per new step).  Stored: the module's state_dict,
the input, the filter outputs that predict_future consumes and everything it returns.
Only data is written.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def build(net, seed, mode="mean", dims=(2, 4, 3)):
    import torch.distributions.multivariate_normal as mvn_mod
    from kvae.sample_control import SampleControl
    from kvae.state_space_model import StateSpaceModel
    torch.manual_seed(seed)
    T, B, steps = 7, 5, 4
    p, n, K = dims
    ssm = StateSpaceModel(a_dim=p, z_dim=n, K=K, dynamics_parameter_network=net, hidden_dim=8,
                          num_layers=2, init_transition_reg_weight=0.9,
                          init_observation_reg_weight=0.9).double().eval()
    with torch.no_grad():
        ssm._mat_Q_L.copy_(0.6 * torch.eye(n) + 0.1 * torch.randn(n, n))
    a = (0.7 * torch.randn(T, B, p)).double()
    sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")
    sc_roll = SampleControl(encoder="mean", decoder="mean", state_transition=mode,
                            observation="mean")
    draws = []
    orig = mvn_mod._standard_normal

    def rec(shape, dtype, device):
        e = orig(shape, dtype=dtype, device=device)
        draws.append(e)
        return e

    with torch.no_grad():
        m, P, mn, Pn, As, Cs, _, _ = ssm.kalman_filter(a, sample_control=sc)
        mvn_mod._standard_normal = rec
        try:
            out = ssm.predict_future(a, m, P, mn, Pn, As, Cs, steps, sc_roll)
        finally:
            mvn_mod._standard_normal = orig
    as_t, means, covs, nmeans, ncovs, As_l, Cs_l = out
    data = {"as_": a, "num_steps": torch.tensor(steps), "filter_means": m, "filter_covariances": P,
            "filter_next_means": mn, "filter_next_covariances": Pn, "mat_As": As, "mat_Cs": Cs,
            "out_as": as_t, "out_means": torch.stack(means), "out_covariances": torch.stack(covs),
            "out_next_means": torch.stack(nmeans), "out_next_covariances": torch.stack(ncovs),
            "out_mat_As": torch.stack(As_l), "out_mat_Cs": torch.stack(Cs_l)}
    if draws:
        assert len(draws) == steps
        data["eps"] = torch.stack([e.reshape(B, n) for e in draws])
    data["dims"] = torch.tensor([p, n, K])
    for k, v in ssm.state_dict().items():
        data["sd/" + k] = v
    return {k: v.detach().cpu().numpy() for k, v in data.items()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    os.makedirs(os.path.join(HERE, "predict"), exist_ok=True)
    for net, seed in (("lstm", 31), ("mlp", 32)):
        np.savez_compressed(os.path.join(HERE, "predict", f"mean_{net}_f64.npz"), **build(net, seed))
        print("wrote", f"predict/mean_{net}_f64.npz")
    # sampled transitions, and dimensions without a kernel of their own (a_dim 3, z_dim 5, K 4)
    for name, net, seed, mode, dims in (("sample_lstm", "lstm", 33, "sample", (2, 4, 3)),
                                        ("sample_mlp", "mlp", 34, "sample", (2, 4, 3)),
                                        ("mean_mlp_n5p3k4", "mlp", 35, "mean", (3, 5, 4)),
                                        ("sample_lstm_n5p3k4", "lstm", 36, "sample", (3, 5, 4))):
        np.savez_compressed(os.path.join(HERE, "predict", f"{name}_f64.npz"),
                            **build(net, seed, mode, dims))
        print("wrote", f"predict/{name}_f64.npz")
