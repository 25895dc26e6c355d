"""kvae_filter_smooth_fused (filter, optional in-loop perceptron network and RTS smoother
as ONE launch with nullable outputs) against the separate calls, which the other GPU tests
pin to the fp64 oracle and to the reference fixtures.

Same statements in both paths, but the compiler contracts multiply-adds differently
where it inlines them into the one-pass kernel, and with the default dispatch the separate
filter call may use its four-lanes-per-sequence form (matrix-core products, another
summation order): tolerance 2e-6 (float32) / 1e-13 (float64) max-norm relative; the
one-tensor and all-tensor variants of the fused call itself must agree bitwise."""
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu

NAMES = ("mean_filt", "cov_filt", "mean_pred", "cov_pred", "mat_As", "mat_Cs",
         "mean_smooth", "cov_smooth", "zs", "as_resampled")


def problem(T, B, n, p, K, dtype, seed, sample):
    torch.manual_seed(seed)
    dev = torch.device("cuda:0")
    r = lambda *s: torch.randn(*s, dtype=dtype, device=dev)
    eye = lambda k: torch.eye(k, dtype=dtype, device=dev)
    d = dict(a=r(T, B, p), mask=(torch.rand(T, B, device=dev) < 0.7).to(dtype),
             A_K=eye(n) * 0.9 + 0.05 * r(K, n, n), C_K=0.3 * r(K, p, n))
    W = 0.2 * r(n, n)
    V = 0.2 * r(p, p)
    d["Q"], d["R"] = 0.5 * eye(n) + W @ W.T, 0.3 * eye(p) + V @ V.T
    d["m0"], d["P0"] = 0.1 * r(n), eye(n)
    d["L_Q"], d["L_R"] = torch.linalg.cholesky(d["Q"]), torch.linalg.cholesky(d["R"])
    if sample:
        d.update(eps_z=r(T, B, n), eps_a=r(T, B, p), eps_state=r(T, B, n), eps_obs=r(T, B, p))
    else:
        d.update(eps_z=None, eps_a=None, eps_state=None, eps_obs=None)
    return d


@pytest.mark.parametrize("lanes", ["1", None])
@pytest.mark.parametrize("sample", [False, True])
@pytest.mark.parametrize("n,p,K,dtype", [(4, 2, 3, torch.float32), (4, 2, 3, torch.float64),
                                         (8, 4, 4, torch.float32), (2, 1, 2, torch.float64),
                                         (5, 3, 4, torch.float64)])
def test_given_weights_equal_separate_calls(n, p, K, dtype, sample, lanes, monkeypatch):
    from kalman_vae_amd import functions as F
    if lanes:
        monkeypatch.setenv("KVAE_SEQ_LANES", lanes)
    T, B = 9, 150
    d = problem(T, B, n, p, K, dtype, n + p, sample)
    alpha = torch.softmax(torch.randn(T + 1, B, K, dtype=dtype, device="cuda:0"), -1)
    with torch.no_grad():
        filt = F.kalman_filter_steps(d["a"], alpha, d["mask"], d["A_K"], d["C_K"], d["Q"], d["R"],
                                     d["m0"], d["P0"])
        smo = F.kalman_smooth_steps(*filt, LR=d["L_R"] if sample else None,
                                    eps_state=d["eps_state"], eps_obs=d["eps_obs"])
    want = dict(zip(NAMES, tuple(filt) + tuple(smo)))
    out = F.filter_smooth_fused(d["a"], d["mask"], d["A_K"], d["C_K"], d["Q"], d["R"], d["m0"],
                                d["P0"], alpha=alpha, L_R=d["L_R"], eps_state=d["eps_state"],
                                eps_obs=d["eps_obs"], want=NAMES)
    assert set(out) == set(NAMES)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for k in NAMES:
        assert rel_err(out[k], want[k]) < tol, (k, rel_err(out[k], want[k]))
    # a consumer that asks for one tensor gets the same values, and nothing else is written
    only = F.filter_smooth_fused(d["a"], d["mask"], d["A_K"], d["C_K"], d["Q"], d["R"], d["m0"],
                                 d["P0"], alpha=alpha, L_R=d["L_R"], eps_state=d["eps_state"],
                                 eps_obs=d["eps_obs"], want=("as_resampled",))
    assert list(only) == ["as_resampled"]
    assert torch.equal(only["as_resampled"], out["as_resampled"])
    assert F.read_status(torch.device("cuda:0")) == 0


@pytest.mark.parametrize("sample", [False, True])
@pytest.mark.parametrize("n,p,K,dtype", [(4, 2, 3, torch.float32), (4, 2, 3, torch.float64),
                                         (3, 2, 2, torch.float64), (5, 3, 4, torch.float32)])
def test_perceptron_in_the_loop_equals_separate_calls(n, p, K, dtype, sample):
    from kalman_vae_amd import functions as F
    T, B, hidden = 11, 130, 16
    d = problem(T, B, n, p, K, dtype, 3 * n + p, sample)
    first = torch.nn.Linear(p, hidden).to(dtype).cuda()
    last = torch.nn.Linear(hidden, K).to(dtype).cuda()
    packed = F.pack_mlp2(first, last)
    m, P, mn, Pn, As, Cs, a_w, alpha = F.masked_filter_mlp(
        d["a"], d["mask"], packed, hidden, d["A_K"], d["C_K"], d["Q"], d["R"], d["m0"], d["P0"],
        d["eps_z"], d["L_Q"], d["eps_a"], d["L_R"])
    with torch.no_grad():
        smo = F.kalman_smooth_steps(m, P, mn, Pn, As, Cs, LR=d["L_R"] if sample else None,
                                    eps_state=d["eps_state"], eps_obs=d["eps_obs"])
    want = dict(zip(NAMES, (m, P, mn, Pn, As, Cs) + tuple(smo)), a_w=a_w, alpha=alpha)
    names = NAMES + ("a_w", "alpha")
    out = F.filter_smooth_fused(d["a"], d["mask"], d["A_K"], d["C_K"], d["Q"], d["R"], d["m0"],
                                d["P0"], packed=packed, hidden=hidden, eps_z=d["eps_z"],
                                L_Q=d["L_Q"], eps_a=d["eps_a"], L_R=d["L_R"],
                                eps_state=d["eps_state"], eps_obs=d["eps_obs"], want=names)
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    for k in names:
        assert rel_err(out[k], want[k]) < tol, (k, rel_err(out[k], want[k]))
    two = F.filter_smooth_fused(d["a"], d["mask"], d["A_K"], d["C_K"], d["Q"], d["R"], d["m0"],
                                d["P0"], packed=packed, hidden=hidden, eps_z=d["eps_z"],
                                L_Q=d["L_Q"], eps_a=d["eps_a"], L_R=d["L_R"],
                                eps_state=d["eps_state"], eps_obs=d["eps_obs"])
    assert sorted(two) == ["a_w", "as_resampled"]
    assert torch.equal(two["a_w"], out["a_w"])
    assert torch.equal(two["as_resampled"], out["as_resampled"])
    assert F.read_status(torch.device("cuda:0")) == 0


@pytest.mark.parametrize("net,masked", [("mlp", True), ("mlp", False), ("lstm", True),
                                        ("lstm", False)])
@pytest.mark.parametrize("mode", ["mean", "sample"])
def test_module_impute_equals_filter_then_smoother(net, masked, mode):
    """StateSpaceModel.impute (one launch where the network allows it) returns the
    filter_as / as_resampled of kalman_filter followed by kalman_smooth, and draws its
    noise in the same order."""
    from kalman_vae_amd import SampleControl, StateSpaceModel
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    ssm = StateSpaceModel(2, 4, 3, net, hidden_dim=16).to(dev)
    ssm.train(mode == "sample")
    T, B = 12, 70
    a = torch.randn(T, B, 2, device=dev)
    mask = (torch.rand(T, B, device=dev) < 0.6).float() if masked else None
    sc = SampleControl(encoder="mean", decoder="mean", state_transition=mode, observation=mode)
    with torch.no_grad():
        torch.manual_seed(11)
        out = ssm.kalman_filter(a, sc, observation_mask=mask)
        smo = ssm.kalman_smooth(a, *out[:6], sample_control=sc)
        torch.manual_seed(11)
        filter_as, as_resampled = ssm.impute(a, sc, observation_mask=mask)
    assert rel_err(filter_as, out[6]) < 2e-6
    assert rel_err(as_resampled, smo[3]) < 2e-6


@pytest.mark.parametrize("masked", [True, False])
@pytest.mark.parametrize("refused", [("filter_smooth_fused",),
                                     ("filter_smooth_fused", "masked_filter_mlp"),
                                     ("filter_smooth_fused", "masked_filter_mlp",
                                      "impute_observation")])
def test_refused_kernel_does_not_draw_twice(masked, refused, monkeypatch):
    """A fused path draws its noise before it learns whether a kernel covers the call.  When
    the kernel refuses (here: made to), the path that runs instead must be served the noise
    already drawn, so that the generator is consumed once and in the reference's order: the
    results and the generator's state afterwards equal those of the plain two calls."""
    from kalman_vae_amd import SampleControl, StateSpaceModel
    from kalman_vae_amd import functions as F
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    ssm = StateSpaceModel(2, 4, 3, "mlp", hidden_dim=16).to(dev)
    ssm.train(True)
    T, B = 9, 40
    a = torch.randn(T, B, 2, device=dev)
    mask = (torch.rand(T, B, device=dev) < 0.6).float() if masked else None
    sc = SampleControl(encoder="mean", decoder="mean", state_transition="sample",
                       observation="sample")
    with torch.no_grad():
        torch.manual_seed(11)
        out = ssm.kalman_filter(a, sc, observation_mask=mask)
        smo = ssm.kalman_smooth(a, *out[:6], sample_control=sc)
        after = torch.randn(3, device=dev)
        for name in refused:
            monkeypatch.setattr(F, name, lambda *a_, **k_: None)
        torch.manual_seed(11)
        filter_as, as_resampled = ssm.impute(a, sc, observation_mask=mask)
        after2 = torch.randn(3, device=dev)
    assert rel_err(filter_as, out[6]) < 2e-6
    assert rel_err(as_resampled, smo[3]) < 2e-6
    assert torch.equal(after, after2)
    assert not ssm._noise_pool and ssm._draw_log is None
