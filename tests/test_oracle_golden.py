"""Pin the CPU oracle to the reference: replay every golden fixture (generated
by running the reference itself, tests/golden/make_golden.py) through
oracle/kalman_oracle.py and compare all outputs and gradients."""
import pytest
import torch

from golden_util import Golden, golden_names, grad_mismatch, rel_err
from oracle import kalman_oracle as O

OUT_MAP = {  # oracle info key -> fixture key
    "filter_means": "filter_means", "filter_covariances": "filter_covariances",
    "filter_next_means": "filter_next_means",
    "filter_next_covariances": "filter_next_covariances",
    "mat_As": "mat_As", "mat_Cs": "mat_Cs", "as_for_weight": "filter_as",
    "weights": "weights", "means": "means", "covariances": "covariances",
    "zs": "zs", "as_resampled": "as_resampled",
    "elbo_obs": "elbo_obs", "elbo_trans": "elbo_trans", "elbo_post": "elbo_post",
}
# the fixture keeps the reference's *weighted* aggregated terms (kalman_vae.py:245-253)
AGG_SIGN = {"agg_obs": 1.0, "agg_trans": 1.0, "agg_post": -1.0}


def run_oracle(g):
    c = g.cfg
    net = g.weight_model()
    a = g["in.as_"].clone().requires_grad_(True)
    A_K = g["sd._mat_A_K"].clone().requires_grad_(True)
    C_K = g["sd._mat_C_K"].clone().requires_grad_(True)
    Q_L = g["sd._mat_Q_L"].clone().requires_grad_(True)
    R_L = g["sd._mat_R_L"].clone().requires_grad_(True)
    n, p = c["n"], c["p"]
    Q = Q_L @ Q_L.T + torch.eye(n, dtype=g.dtype) * float(g["in.Q_reg"])
    R = R_L @ R_L.T + torch.eye(p, dtype=g.dtype) * float(g["in.R_reg"])
    A_use, C_use = A_K, C_K
    if c.get("fix_matrices", False):
        Q, R, A_use, C_use = Q.detach(), R.detach(), A_K.detach(), C_K.detach()
    if not c.get("learn_noise_covariance", False):
        Q, R = Q.detach(), R.detach()
    net.clear_hidden_state()
    mask = g["in.mask"] if g.has("in.mask") else None
    obj, info = O.kalman_objective(
        a, A_use, C_use, Q, R, g["in.m0"], g["in.P0"], net, g["eps.elbo"], mask=mask,
        filter_noise=g.noise("filter"), smooth_noise=g.noise("smooth"),
        learn_weight_model=c.get("learn_weight_model", True),
        symmetrize=c.get("symmetrize", True), burn_in=c.get("burn_in", 0),
        sequence_operation=c.get("sequence_operation", "mean"),
        batch_operation=c.get("batch_operation", "mean"),
        kalman_weight=c.get("kalman_weight", 1.0))
    params = {"as_": a, "_mat_A_K": A_K, "_mat_C_K": C_K, "_mat_Q_L": Q_L, "_mat_R_L": R_L}
    for k, v in net.named_parameters():
        params["weight_model." + k] = v
    return obj, info, params


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name):
    g = Golden(name)
    tol = 1e-10 if g.dtype == torch.float64 else 2e-4
    if g.stored_f32:
        tol = 2e-6  # float64 run stored as float32
    obj, info, params = run_oracle(g)
    assert torch.equal(g["in.Q"], g["in.Q"].T.T)
    for ok, gk in OUT_MAP.items():
        assert rel_err(info[ok], g["out." + gk]) < tol, (ok, rel_err(info[ok], g["out." + gk]))
    kw = c_kw = g.cfg.get("kalman_weight", 1.0)
    for k, sign in AGG_SIGN.items():
        assert rel_err(sign * kw * info[k], g["out." + k]) < tol, k
    assert rel_err(obj, g["out.objective"]) < tol
    loss = obj
    for ok, gk in OUT_MAP.items():
        if (g.has("cot." + gk) and info[ok].requires_grad
                and not g.cfg.get("objective_only", False)):
            loss = loss + (g["cot." + gk] * info[ok]).sum()
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    for (k, _), gv in zip(params.items(), grads):
        if g.has("grad." + k) or g.has("gradsub." + k):
            assert gv is not None, k
            assert grad_mismatch(g, k, gv, tol) < tol * 10, (k, grad_mismatch(g, k, gv, tol))
        else:
            assert gv is None or float(gv.abs().max()) == 0.0, k
