"""The module at the benchmark's full sizes.  The oracle cannot run 4096 x 20 or
1024 x 1000 sequence-steps in test time, so these tests use what the domain
offers: sequences are independent given the parameters, hence

  * a random SUBSET of the sequences, pushed through the float64 oracle alone
    with the same parameters and the same noise, must reproduce the full-size
    run's per-sequence outputs, per-(t,b) ELBO terms and the gradient w.r.t.
    those sequences' observations (float32 tolerance 1e-4, max-norm relative
    per tensor);
  * permuting the batch permutes every per-sequence output bit for bit;
  * the objective with sum/sum aggregation equals the sum of the per-(t,b) terms.

Shapes: configuration 2 (T=20, B=4096, z=4, a=2, K=3, LSTM weights, training
mode), configuration 4 (T=1000, B=1024, z=32, a=16, K=8), configuration 3
(masked imputation, one 65536-sequence shard, inference mode).
"""
import copy

import pytest
import torch

from golden_util import rel_err
from oracle import kalman_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def _model(n, p, K, net, seed, train):
    from kalman_vae_amd import StateSpaceModel
    torch.manual_seed(seed)
    ssm = StateSpaceModel(a_dim=p, z_dim=n, K=K, dynamics_parameter_network=net)
    with torch.no_grad():      # generic noise covariances, not multiples of the identity
        ssm._mat_Q_L.copy_(0.5 * torch.eye(n) + 0.05 * torch.randn(n, n))
        ssm._mat_R_L.copy_(0.4 * torch.eye(p) + 0.05 * torch.randn(p, p))
    ssm.rng_parity = False     # one noise tensor per call: (T,B,.) tensors to hand to the oracle
    return ssm.to(DEV).train(train)


def _record_draws(monkeypatch):
    import kalman_vae_amd.elbo as ke
    import kalman_vae_amd.state_space_model as sm
    draws, orig = [], sm._standard_normal

    def rec(shape, dtype, device):
        e = orig(shape, dtype=dtype, device=device)
        draws.append(e)
        return e

    monkeypatch.setattr(sm, "_standard_normal", rec)
    monkeypatch.setattr(ke, "_standard_normal", rec)
    return draws


def _oracle_subset(ssm, a, idx, smooth_noise, eps_elbo, mask=None, burn_in=0):
    """float64 oracle on sequences idx with the module's parameters; returns
    (objective, info, a64) with sum/sum aggregation."""
    cpu = copy.deepcopy(ssm).cpu().double()
    cpu.weight_model.clear_hidden_state()
    pick = lambda t: None if t is None else t[:, idx].detach().cpu().double()
    a64 = pick(a).requires_grad_(True)
    obj, info = O.kalman_objective(
        a64, cpu._mat_A_K, cpu._mat_C_K, cpu.mat_Q, cpu.mat_R, cpu.initial_state_mean,
        cpu.initial_state_covariance, cpu.weight_model, pick(eps_elbo), mask=pick(mask),
        smooth_noise=(pick(smooth_noise[0]), pick(smooth_noise[1])), burn_in=burn_in,
        sequence_operation="sum", batch_operation="sum")
    return obj, info, a64


PER_SEQUENCE = ["filter_means", "filter_covariances", "filter_next_means", "filter_next_covariances",
                "means", "covariances", "zs", "as_resampled"]


def _compare_subset(info, oinfo, idx, failures, keys=PER_SEQUENCE):
    for k in keys:
        e = rel_err(info[k][:, idx], oinfo[k])
        if not e < TOL:
            failures.append(f"{k}: {e:.2e}")


def _training_run(ssm, a, burn_in, monkeypatch):
    from kalman_vae_amd import SampleControl, kalman_elbo
    draws = _record_draws(monkeypatch)
    a = a.clone().requires_grad_(True)
    obj, info = kalman_elbo(ssm, a, SampleControl(), burn_in=burn_in, sequence_operation="sum",
                            batch_operation="sum")
    (g_a,) = torch.autograd.grad(obj, [a])
    # one launch draws all the noise of the step: smoother state, smoother observation, ELBO
    assert len(draws) == 1, [tuple(d.shape) for d in draws]
    T, B = a.shape[:2]
    sizes = [T * B * ssm.z_dim, T * B * ssm.a_dim, T * B * ssm.z_dim]
    return obj, info, g_a, [piece.view(T, B, -1) for piece in draws[0].split(sizes)]


def _check_against_oracle(ssm, a, obj, info, g_a, draws, idx, burn_in):
    failures = []
    oobj, oinfo, a64 = _oracle_subset(ssm, a, idx, (draws[0], draws[1]), draws[2], burn_in=burn_in)
    _compare_subset(info, oinfo, idx, failures)
    obs, trans, post = info["elbo_terms"]
    for name, got in (("obs", obs), ("trans", trans), ("post", post)):
        e = rel_err(got[:, idx], oinfo["elbo_" + name])
        if not e < TOL:
            failures.append(f"elbo_{name}: {e:.2e}")
    (og,) = torch.autograd.grad(oobj, [a64])
    e = rel_err(g_a[:, idx], og)
    if not e < TOL:
        failures.append(f"d objective / d a: {e:.2e}")
    total = (obs.double().sum() + trans.double().sum() - post.double().sum())
    e = abs(float(obj.detach()) - float(total)) / abs(float(total))
    if not e < 1e-5:
        failures.append(f"objective vs sum of terms: {e:.2e}")
    assert not failures, "; ".join(failures)


def test_config2_step_as_benchmarked(monkeypatch):
    """Exactly the step bench.py times by default: the fused objective (value and gradient in
    one call, per-pair terms kept), captured and replayed as ONE HIP graph, T=20, B=4096, the
    reference's default network (2 x 128 LSTM) on split-bf16 products.  A subset of sequences
    through the float64 oracle must give the same per-pair terms and the same gradient
    w.r.t. the observations; the objective must equal the sum of the terms; and every
    parameter gradient must agree with the same step on exact fp32 products."""
    import kalman_vae_amd.elbo as ke
    from kalman_vae_amd import SampleControl
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd.graphed import GraphedElboStep
    T, B, n, p, K, burn_in = 20, 4096, 4, 2, 3, 3
    ssm = _model(n, p, K, "lstm", seed=5, train=True)
    assert ssm.weight_model.hidden_dim == 128 and ssm.weight_model.num_layers == 2
    ssm.fused_loss = ssm.fused_loss_terms = True
    g = torch.Generator().manual_seed(16)
    a = (0.7 * torch.randn(T, B, p, generator=g)).to(DEV)
    eps = torch.randn(T, B, n, generator=g).to(DEV)     # the ELBO sample's noise, fixed

    def fixed(shape, dtype, device):
        assert tuple(shape) == (T, B, n)
        return eps

    monkeypatch.setattr(ke, "_standard_normal", fixed)
    lib = capi.load()
    assert lib.kvae_lstm128_get_precision() == 1
    kw = dict(burn_in=burn_in, sequence_operation="sum", batch_operation="sum")
    step = GraphedElboStep(ssm, a, SampleControl(), **kw)
    obj = step(a)
    torch.cuda.synchronize()
    assert step.info["filter_means"] is None            # the fused path: nothing materialised
    obs, trans, post = step.info["elbo_terms"]
    grads = {k: v.grad.detach().clone() for k, v in ssm.named_parameters() if v.grad is not None}
    g_a = step.as_grad.detach().clone()
    failures = []
    idx = torch.randperm(B, generator=g)[:40]
    zeros = (torch.zeros(T, B, n, device=DEV), torch.zeros(T, B, p, device=DEV))
    oobj, oinfo, a64 = _oracle_subset(ssm, a, idx, zeros, eps, burn_in=burn_in)
    for name, got in (("obs", obs), ("trans", trans), ("post", post)):
        e = rel_err(got[:, idx], oinfo["elbo_" + name])
        if not e < TOL:
            failures.append(f"elbo_{name}: {e:.2e}")
    (og,) = torch.autograd.grad(oobj, [a64])
    e = rel_err(g_a[:, idx], -og)                        # the graph differentiates -objective
    if not e < TOL:
        failures.append(f"d(-objective)/d a: {e:.2e}")
    total = obs.double().sum() + trans.double().sum() - post.double().sum()
    e = abs(float(obj) - float(total)) / abs(float(total))
    if not e < 1e-5:
        failures.append(f"objective vs sum of terms: {e:.2e}")
    # a second replay gives the same numbers (fixed noise): nothing is left over between replays
    obj_again = float(step(a))
    torch.cuda.synchronize()
    if obj_again != float(obj):
        failures.append(f"replay changed the objective: {obj_again} vs {float(obj)}")
    # the same step with every weight-network product on fp32 matrix instructions
    lib.kvae_lstm128_set_precision(0)
    try:
        exact = GraphedElboStep(ssm, a, SampleControl(), **kw)
        obj32 = exact(a)
        torch.cuda.synchronize()
        e = abs(float(obj32) - float(obj)) / abs(float(obj))
        if not e < 1e-5:
            failures.append(f"objective, split vs exact products: {e:.2e}")
        for k, v in ssm.named_parameters():
            if v.grad is None:
                continue
            e = rel_err(grads[k], v.grad)
            if not e < TOL:
                failures.append(f"d/d {k}, split vs exact products: {e:.2e}")
        e = rel_err(g_a, exact.as_grad)
        if not e < TOL:
            failures.append(f"d/d a, split vs exact products: {e:.2e}")
    finally:
        lib.kvae_lstm128_set_precision(1)
    assert not failures, "; ".join(failures)


def test_config2_full_batch_subset_and_permutation(monkeypatch):
    T, B, n, p, K, burn_in = 20, 4096, 4, 2, 3, 3
    ssm = _model(n, p, K, "lstm", seed=5, train=True)
    g = torch.Generator().manual_seed(6)
    a = (0.7 * torch.randn(T, B, p, generator=g)).to(DEV)
    obj, info, g_a, draws = _training_run(ssm, a, burn_in, monkeypatch)
    idx = torch.randperm(B, generator=g)[:40]
    _check_against_oracle(ssm, a, obj, info, g_a, draws, idx, burn_in)

    # the same batch in another order, with the same noise per sequence
    import kalman_vae_amd.elbo as ke
    import kalman_vae_amd.state_space_model as sm
    perm = torch.randperm(B, generator=g).to(DEV)
    replay = torch.cat([d[:, perm].reshape(-1) for d in draws])

    def fake(shape, dtype, device):
        assert tuple(shape) == tuple(replay.shape)
        return replay

    monkeypatch.setattr(sm, "_standard_normal", fake)
    monkeypatch.setattr(ke, "_standard_normal", fake)
    from kalman_vae_amd import SampleControl, kalman_elbo
    a2 = a[:, perm].clone().requires_grad_(True)
    obj2, info2 = kalman_elbo(ssm, a2, SampleControl(), burn_in=burn_in, sequence_operation="sum",
                              batch_operation="sum")
    (g2,) = torch.autograd.grad(obj2, [a2])
    for k in PER_SEQUENCE + ["mat_As", "mat_Cs", "weights"]:
        assert torch.equal(info2[k], info[k][:, perm]), k
    assert torch.equal(g2, g_a[:, perm])
    assert abs(float(obj2.detach()) - float(obj.detach())) < 1e-5 * abs(float(obj.detach()))


def test_config4_long_large_state_subset(monkeypatch):
    T, B, n, p, K, burn_in = 1000, 1024, 32, 16, 8, 3
    ssm = _model(n, p, K, "lstm", seed=7, train=True)
    g = torch.Generator().manual_seed(8)
    a = (0.7 * torch.randn(T, B, p, generator=g)).to(DEV)
    obj, info, g_a, draws = _training_run(ssm, a, burn_in, monkeypatch)
    idx = torch.tensor([0, 517, 1023])
    _check_against_oracle(ssm, a, obj, info, g_a, draws, idx, burn_in)


@pytest.mark.parametrize("net", ["lstm", "mlp"])
def test_config3_masked_imputation_shard_subset(net, monkeypatch):
    from kalman_vae_amd import SampleControl
    T, B, n, p, K = 20, 65536, 4, 2, 3
    ssm = _model(n, p, K, net, seed=9, train=False)
    g = torch.Generator().manual_seed(10)
    a = (0.7 * torch.randn(T, B, p, generator=g)).to(DEV)
    mask = (torch.rand(T, B, generator=g) > 0.3).float()
    mask[:3] = 1.0
    mask = mask.to(DEV)
    sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")
    with torch.no_grad():
        m, P, mn, Pn, As, Cs, filter_as, weights = ssm.kalman_filter(
            a, sample_control=sc, observation_mask=mask)
        ms, Ps, zs, ares = ssm.kalman_smooth(
            a, filter_means=m, filter_covariances=P, filter_next_means=mn,
            filter_next_covariances=Pn, mat_As=As, mat_Cs=Cs, sample_control=sc)
    idx = torch.randperm(B, generator=g)[:32]
    cpu = copy.deepcopy(ssm).cpu().double()
    cpu.weight_model.clear_hidden_state()
    pick = lambda t: t[:, idx].cpu().double()
    with torch.no_grad():
        f = O.kalman_filter(pick(a), cpu._mat_A_K, cpu._mat_C_K, cpu.mat_Q, cpu.mat_R,
                            cpu.initial_state_mean, cpu.initial_state_covariance, cpu.weight_model,
                            mask=pick(mask))
        s = O.kalman_smooth(f["filter_means"], f["filter_covariances"], f["filter_next_means"],
                            f["filter_next_covariances"], f["mat_As"], f["mat_Cs"], cpu.mat_R)
    failures = []
    got = {"filter_means": m, "filter_covariances": P, "filter_next_means": mn,
           "filter_next_covariances": Pn, "filter_as": filter_as, "means": ms, "covariances": Ps,
           "as_resampled": ares}
    for k, v in got.items():
        want = f["as_for_weight"] if k == "filter_as" else (f[k] if k in f else s[k])
        e = rel_err(v[:, idx], want)
        if not e < TOL:
            failures.append(f"{k}: {e:.2e}")
    # where nothing was missing the imputed sequence is the observed one
    seen = mask.bool().unsqueeze(-1).expand_as(a)
    assert torch.equal(filter_as[seen], a[seen])
    # the imputation consumer's call (one launch with the perceptron network, the two calls
    # otherwise) against the same oracle subset
    imp_as, imp_res = ssm.impute(a, sc, observation_mask=mask)
    for k, v, want in (("impute.filter_as", imp_as, f["as_for_weight"]),
                       ("impute.as_resampled", imp_res, s["as_resampled"])):
        e = rel_err(v[:, idx], want)
        if not e < TOL:
            failures.append(f"{k}: {e:.2e}")
    assert not failures, "; ".join(failures)
