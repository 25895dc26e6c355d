"""CPU check of the device math: the per-sequence source that the HIP kernels
execute (kalman_vae_amd/csrc/kalman_core.h) is compiled for the host by
tests/hostsim and compared, forward and adjoint, with the oracle.  This is how
the hand-derived adjoints are validated in a container without a GPU; the GPU
tests (test_gpu_kernels.py) repeat the same checks on the HIP library."""
import pytest
import torch

import kernel_checks as KC
from kernel_harness import Backend

DIMS = [(4, 2, 3), (2, 1, 2), (3, 2, 2), (8, 4, 4), (4, 2, 8)]


@pytest.fixture(scope="module")
def be():
    return Backend("hostsim")


@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("mask_kind", ["none", "binary", "fractional"])
def test_filter_f64(be, n, p, K, mask_kind):
    KC.check_filter(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + p,
                    mask_kind=mask_kind)


@pytest.mark.parametrize("symmetrize,burn_in,carry", [(False, 0, False), (True, 2, False),
                                                      (True, 0, True), (False, 3, True)])
def test_filter_options(be, symmetrize, burn_in, carry):
    KC.check_filter(be, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=5, mask_kind="binary",
                    symmetrize=symmetrize, burn_in=burn_in, per_sequence_carry=carry)


def test_filter_f32(be):
    KC.check_filter(be, T=20, B=5, n=4, p=2, K=3, dtype=torch.float32, seed=9, mask_kind="binary")


def test_filter_single_step(be):
    KC.check_filter(be, T=1, B=4, n=4, p=2, K=3, dtype=torch.float64, seed=3, mask_kind="binary",
                    per_sequence_carry=True)


@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("sample", [False, True])
def test_smooth_f64(be, n, p, K, sample):
    KC.check_smooth(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n, sample=sample)


@pytest.mark.parametrize("symmetrize,burn_in", [(False, 0), (True, 3), (False, 2)])
def test_smooth_options(be, symmetrize, burn_in):
    KC.check_smooth(be, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=11, sample=True,
                    symmetrize=symmetrize, burn_in=burn_in)


def test_smooth_f32_and_T1(be):
    KC.check_smooth(be, T=20, B=4, n=4, p=2, K=3, dtype=torch.float32, seed=2, sample=True)
    KC.check_smooth(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=2, sample=True)


@pytest.mark.parametrize("n,p,K", DIMS)
def test_elbo_f64(be, n, p, K):
    KC.check_elbo(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + 1)


def test_elbo_f32_and_T1(be):
    KC.check_elbo(be, T=20, B=4, n=4, p=2, K=3, dtype=torch.float32, seed=4)
    KC.check_elbo(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=4)


# -- fused objective (kalman_loss.h): filter -> smoother -> terms in one pass ------
@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("mask_kind", ["none", "fractional"])
def test_loss_f64(be, n, p, K, mask_kind):
    KC.check_loss(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + 2,
                  mask_kind=mask_kind)


@pytest.mark.parametrize("symmetrize,burn_in,stride", [(False, 0, 1), (True, 3, 0), (False, 2, 0),
                                                       (True, 9, 1)])
def test_loss_options(be, symmetrize, burn_in, stride):
    KC.check_loss(be, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=13, mask_kind="binary",
                  symmetrize=symmetrize, burn_in=burn_in, stride=stride)


def test_loss_f32_and_short_sequences(be):
    KC.check_loss(be, T=20, B=5, n=4, p=2, K=3, dtype=torch.float32, seed=6, burn_in=3)
    KC.check_loss(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=6)
    KC.check_loss(be, T=2, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=7, burn_in=1)


# -- the same objective with one sequence over 16 lanes (kalman_row.h): the host build runs the
# -- device source on an array of 16 lanes whose exchanges are permutations -------------
@pytest.fixture(scope="module")
def br():
    return Backend("hostsim_row")


@pytest.mark.parametrize("n,p,K", [(4, 2, 3), (2, 1, 2), (3, 2, 2), (4, 2, 8)])
@pytest.mark.parametrize("mask_kind", ["none", "fractional"])
def test_row_loss_f64(br, n, p, K, mask_kind):
    KC.check_loss(br, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + 2,
                  mask_kind=mask_kind)


@pytest.mark.parametrize("symmetrize,burn_in,stride", [(False, 0, 1), (True, 3, 0), (False, 2, 0),
                                                       (True, 9, 1)])
def test_row_loss_options(br, symmetrize, burn_in, stride):
    KC.check_loss(br, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=13, mask_kind="binary",
                  symmetrize=symmetrize, burn_in=burn_in, stride=stride)


def test_row_loss_f32_and_short_sequences(br):
    KC.check_loss(br, T=20, B=5, n=4, p=2, K=3, dtype=torch.float32, seed=6, burn_in=3)
    KC.check_loss(br, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=6)
    KC.check_loss(br, T=2, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=7, burn_in=1)


# -- wavefront-per-sequence source (kalman_wave.h), same checks ----------------
WAVE_DIMS = [(4, 2, 3), (6, 3, 2), (16, 8, 4), (32, 16, 8)]


@pytest.fixture(scope="module")
def bw():
    return Backend("hostsim_wave")


@pytest.mark.parametrize("n,p,K", WAVE_DIMS)
@pytest.mark.parametrize("mask_kind", ["none", "fractional"])
def test_wave_filter_f64(bw, n, p, K, mask_kind):
    KC.check_filter(bw, T=5, B=2, n=n, p=p, K=K, dtype=torch.float64, seed=n + p,
                    mask_kind=mask_kind, burn_in=1)


def test_wave_filter_options(bw):
    KC.check_filter(bw, T=6, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=5, mask_kind="binary",
                    symmetrize=False, burn_in=2, per_sequence_carry=True)
    KC.check_filter(bw, T=1, B=2, n=16, p=8, K=4, dtype=torch.float64, seed=6, mask_kind="binary",
                    per_sequence_carry=True)


@pytest.mark.parametrize("n,p,K", WAVE_DIMS)
@pytest.mark.parametrize("sample", [False, True])
def test_wave_smooth_f64(bw, n, p, K, sample):
    KC.check_smooth(bw, T=5, B=2, n=n, p=p, K=K, dtype=torch.float64, seed=n, sample=sample,
                    burn_in=1)


def test_wave_smooth_options(bw):
    KC.check_smooth(bw, T=6, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=11, sample=True,
                    symmetrize=False, burn_in=2)
    KC.check_smooth(bw, T=1, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=12, sample=True)


@pytest.mark.parametrize("n,p,K", WAVE_DIMS)
def test_wave_elbo_f64(bw, n, p, K):
    KC.check_elbo(bw, T=5, B=2, n=n, p=p, K=K, dtype=torch.float64, seed=n + 1)


def test_wave_f32_large_state(bw):
    KC.check_filter(bw, T=8, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=1, mask_kind="binary")
    KC.check_smooth(bw, T=8, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=2, sample=True)
    KC.check_elbo(bw, T=8, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=3)
    KC.check_elbo(bw, T=1, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=4)


# -- masked filter with the two-layer perceptron network inside the loop -----------
@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("sample", [False, True])
def test_masked_filter_with_network_in_the_loop(n, p, K, sample):
    """filter_forward_masked_mlp_sequence (the source of kvae_filter_forward_masked_mlp)
    against the oracle's masked filter with the same perceptron as its weight function."""
    import ctypes as C
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd.functions import pack_mlp2
    from kernel_harness import hostsim_lib
    from oracle import kalman_oracle as O
    torch.manual_seed(n * 7 + p)
    T, B, H, dt = 7, 5, 11, torch.float64
    first, last = torch.nn.Linear(p, H).to(dt), torch.nn.Linear(H, K).to(dt)
    A_K = (torch.eye(n) + 0.2 * torch.randn(K, n, n)).to(dt)
    C_K = (0.5 * torch.randn(K, p, n)).to(dt)
    Lq, Lr = 0.3 * torch.eye(n) + 0.05 * torch.randn(n, n), 0.3 * torch.eye(p) + 0.05 * torch.randn(p, p)
    Q, R = (Lq @ Lq.T + 0.01 * torch.eye(n)).to(dt), (Lr @ Lr.T + 0.01 * torch.eye(p)).to(dt)
    m0, P0 = torch.randn(n).to(dt), (2.0 * torch.eye(n)).to(dt)
    a = torch.randn(T, B, p).to(dt)
    mask = (torch.rand(T, B) < 0.6).to(dt)
    eps_z = torch.randn(T, B, n).to(dt) if sample else None
    eps_a = torch.randn(T, B, p).to(dt) if sample else None
    with torch.no_grad():
        net = lambda x: torch.softmax(last(torch.relu(first(x))), dim=-1)
        want = O.kalman_filter(a, A_K, C_K, Q, R, m0, P0, net, mask=mask, eps_state=eps_z,
                               eps_obs=eps_a)
    new = lambda *s: torch.zeros(s, dtype=dt)
    m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
    As, Cs, a_w, alpha = new(T + 1, B, n, n), new(T + 1, B, p, n), new(T, B, p), new(T + 1, B, K)
    alpha[0] = 1.0 / K
    status = torch.zeros(1, dtype=torch.int32)
    L_Q = torch.linalg.cholesky(Q).contiguous() if sample else None
    L_R = torch.linalg.cholesky(R).contiguous() if sample else None
    packed = pack_mlp2(first, last)
    dims = capi.Dims(T, B, n, p, K, capi.KVAE_F64, 1, 0)
    io = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                   m_in=m0, P_in=P0, carry_per_sequence=0, mean_filt=m, cov_filt=P, mean_pred=mn,
                   cov_pred=Pn, mat_As=As, mat_Cs=Cs, status=status)
    q = capi.fill(capi.MaskedMlp(), hidden=H, packed=packed, eps_z=eps_z, L_Q=L_Q, eps_a=eps_a,
                  L_R=L_R, a_w=a_w)
    fn = hostsim_lib().hostsim_filter_forward_masked_mlp
    fn.argtypes = [C.POINTER(capi.Dims), C.POINTER(capi.FilterFwd), C.POINTER(capi.MaskedMlp)]
    fn.restype = C.c_int
    assert fn(C.byref(dims), C.byref(io), C.byref(q)) == 0
    assert int(status[0]) == 0
    got = {"filter_means": m, "filter_covariances": P, "filter_next_means": mn,
           "filter_next_covariances": Pn, "mat_As": As, "mat_Cs": Cs, "as_for_weight": a_w,
           "weights": alpha[:T]}
    for key, val in got.items():
        assert KC.rel_err(val, want[key]) < 1e-11, key


@pytest.mark.parametrize("case", ["mean_mlp", "sample_mlp"])
def test_rollout_statements_match_reference(case):
    """The rollout statements the HIP kernel compiles (kalman_rollout.h, built for the host)
    against outputs of the reference's predict_future (tests/golden/predict/*.npz, made by
    tests/golden/make_golden_predict.py), float64, 1e-10: all new steps in one call with the
    two-layer perceptron evaluated in the loop."""
    import ctypes as C
    import os
    import numpy as np
    from kalman_vae_amd import _capi as capi
    from kernel_harness import hostsim_lib
    here = os.path.dirname(os.path.abspath(__file__))
    d = {k: torch.from_numpy(v) for k, v in
         np.load(os.path.join(here, "golden", "predict", f"{case}_f64.npz")).items()}
    p, n, K = (int(v) for v in d["dims"])
    S, T, B = int(d["num_steps"]), d["as_"].shape[0], d["as_"].shape[1]
    sd = {k[3:]: v for k, v in d.items() if k.startswith("sd/")}
    W1, b1 = sd["weight_model.layers.0.weight"], sd["weight_model.layers.0.bias"]
    W2, b2 = sd["weight_model.layers.1.weight"], sd["weight_model.layers.1.bias"]
    hidden = W1.shape[0]
    packed = torch.cat([torch.cat([W1, b1.unsqueeze(1), W2.t()], dim=1).reshape(-1), b2]).contiguous()
    Q_L = sd["_mat_Q_L"]
    Q = (Q_L @ Q_L.T + 1e-3 * torch.eye(n, dtype=torch.float64)).contiguous()
    new = lambda *s: torch.zeros(*s, dtype=torch.float64)
    out = dict(as_new=new(S, B, p), means=new(S, B, n), covs=new(S, B, n, n),
               mean_pred_new=new(S, B, n), cov_pred_new=new(S, B, n, n),
               mat_As_new=new(S, B, n, n), mat_Cs_new=new(S, B, p, n))
    status = torch.zeros(1, dtype=torch.int32)
    keep = dict(A_K=sd["_mat_A_K"].contiguous(), C_K=sd["_mat_C_K"].contiguous(), Q=Q,
                mean_pred=d["filter_next_means"][-1].contiguous(),
                cov_pred=d["filter_next_covariances"][-1].contiguous(),
                mat_C=d["mat_Cs"][T].contiguous(), packed_mlp=packed)
    if "eps" in d:
        keep["eps"] = d["eps"].contiguous()
    io = capi.fill(capi.Rollout(), hidden=hidden, phase=capi.ROLLOUT_ALL,
                   sample=int("eps" in d), status=status, **keep, **out)
    dims = capi.Dims(S, B, n, p, K, capi.KVAE_F64, 1, 0)
    lib = hostsim_lib()
    lib.hostsim_predict_future.argtypes = [C.POINTER(capi.Dims), C.POINTER(capi.Rollout)]
    assert lib.hostsim_predict_future(C.byref(dims), C.byref(io)) == 0
    assert int(status) == 0
    for got, key, skip in ((out["as_new"], "out_as", T), (out["means"], "out_means", T),
                           (out["covs"], "out_covariances", T),
                           (out["mean_pred_new"], "out_next_means", T),
                           (out["cov_pred_new"], "out_next_covariances", T),
                           (out["mat_As_new"], "out_mat_As", T + 1),
                           (out["mat_Cs_new"], "out_mat_Cs", T + 1)):
        want = d[key][skip:]
        assert got.shape == want.shape, key
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 1e-10, (key, err)


@pytest.mark.parametrize("n,p,K", [(4, 2, 3), (3, 2, 2)])
def test_fused_filter_smoother_statements_equal_separate_calls(n, p, K):
    """The one-pass filter + smoother (kalman_filter_smooth.h, built for the host) against
    the separate filter and smoother statements (pinned to the oracle by the tests above):
    bitwise equal in float64, sampled modes, masked, every output requested."""
    import ctypes as C
    from kalman_vae_amd import _capi as capi
    from kernel_harness import Backend, hostsim_lib
    torch.manual_seed(n)
    T, B, dt = 6, 5, torch.float64
    r = lambda *s: torch.randn(*s, dtype=dt)
    a, alpha = r(T, B, p), torch.softmax(r(T + 1, B, K), -1).contiguous()
    mask = (torch.rand(T, B) < 0.7).to(dt)
    A_K = (torch.eye(n, dtype=dt) * 0.9 + 0.05 * r(K, n, n)).contiguous()
    C_K = 0.3 * r(K, p, n)
    Q, R = torch.eye(n, dtype=dt) * 0.5, torch.eye(p, dtype=dt) * 0.3
    m0, P0 = 0.1 * r(n), torch.eye(n, dtype=dt)
    LR = torch.linalg.cholesky(R).contiguous()
    es, eo = r(T, B, n), r(T, B, p)
    be = Backend("hostsim")
    f = be.filter_forward(a, alpha, mask, A_K, C_K, Q, R, m0, P0)
    s = be.smooth_forward(f["mean_filt"], f["cov_filt"], f["mean_pred"], f["cov_pred"],
                          f["mat_As"], f["mat_Cs"], LR, es, eo)
    names_f = ("mean_filt", "cov_filt", "mean_pred", "cov_pred", "mat_As", "mat_Cs")
    names_s = ("mean_smooth", "cov_smooth", "zs", "as_resampled")
    out_f = {k: torch.full_like(f[k], float("nan")) for k in names_f}
    out_s = {k: torch.full_like(s[k], float("nan")) for k in names_s}
    st = torch.zeros(1, dtype=torch.int32)
    fio = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                    m_in=m0, P_in=P0, carry_per_sequence=0, status=st, **out_f)
    sio = capi.fill(capi.SmoothFwd(), LR=LR, eps_state=es, eps_obs=eo, status=st, **out_s)
    lib = hostsim_lib()
    lib.hostsim_filter_smooth_fused.argtypes = [C.POINTER(capi.Dims), C.POINTER(capi.FilterFwd),
                                                C.c_void_p, C.POINTER(capi.SmoothFwd)]
    d = be.dims(T, B, n, p, K, dt)
    assert lib.hostsim_filter_smooth_fused(C.byref(d), C.byref(fio), None, C.byref(sio)) == 0
    for k in names_f:
        assert torch.equal(out_f[k], f[k]), k
    for k in names_s:
        assert torch.equal(out_s[k], s[k]), k


@pytest.mark.parametrize("n,p,K", [(4, 2, 3), (3, 2, 2)])
@pytest.mark.parametrize("sample,burn_in,learn", [(False, 0, True), (True, 2, True),
                                                  (True, 0, False)])
def test_masked_filter_with_network_adjoint(n, p, K, sample, burn_in, learn):
    """filter_backward_masked_mlp_sequence (the source of kvae_filter_backward_masked_mlp)
    against autograd through the oracle's masked filter with the same perceptron: random
    cotangents on all eight outputs, gradients w.r.t. observations, bases, noise
    covariances (through the imputation's Cholesky factors too), initial state and the four
    parameter tensors of the network.  float64, 1e-9."""
    import ctypes as C
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd.functions import pack_mlp2
    from kernel_harness import hostsim_lib
    from oracle import kalman_oracle as O
    torch.manual_seed(n * 11 + p + burn_in)
    T, B, H, dt = 6, 4, 7, torch.float64
    first, last = torch.nn.Linear(p, H).to(dt), torch.nn.Linear(H, K).to(dt)
    leaf = lambda t: t.to(dt).requires_grad_(True)
    A_K = leaf(torch.eye(n) + 0.2 * torch.randn(K, n, n))
    C_K = leaf(0.5 * torch.randn(K, p, n))
    Lq = 0.3 * torch.eye(n) + 0.05 * torch.randn(n, n)
    Lr = 0.3 * torch.eye(p) + 0.05 * torch.randn(p, p)
    Q, R = leaf(Lq @ Lq.T + 0.01 * torch.eye(n)), leaf(Lr @ Lr.T + 0.01 * torch.eye(p))
    m0, P0 = leaf(torch.randn(n)), leaf(2.0 * torch.eye(n) + 0.1 * torch.ones(n, n))
    a = leaf(torch.randn(T, B, p))
    mask = torch.rand(T, B).to(dt)          # fractional: the blend is arithmetic
    mask[0] = 1.0
    eps_z = torch.randn(T, B, n).to(dt) if sample else None
    eps_a = torch.randn(T, B, p).to(dt) if sample else None
    net = lambda x: torch.softmax(last(torch.relu(first(x))), dim=-1)
    want = O.kalman_filter(a, A_K, C_K, Q, R, m0, P0, net, mask=mask, eps_state=eps_z,
                           eps_obs=eps_a, learn_weight_model=learn, burn_in=burn_in)
    names = ["filter_means", "filter_covariances", "filter_next_means", "filter_next_covariances",
             "mat_As", "mat_Cs", "as_for_weight", "weights"]
    cot = {k: torch.randn_like(want[k]) for k in names}
    loss = sum((cot[k] * want[k]).sum() for k in names)
    params = [a, A_K, C_K, Q, R, m0, P0, first.weight, first.bias, last.weight, last.bias]
    gref = torch.autograd.grad(loss, params, allow_unused=True)

    # forward through the host build, then its adjoint
    new = lambda *s: torch.zeros(s, dtype=dt)
    c = lambda t: t.detach().contiguous()
    m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
    As, Cs, a_w, alpha = new(T + 1, B, n, n), new(T + 1, B, p, n), new(T, B, p), new(T + 1, B, K)
    alpha[0] = 1.0 / K
    status = torch.zeros(1, dtype=torch.int32)
    L_Q = torch.linalg.cholesky(c(Q)).contiguous().requires_grad_(True) if sample else None
    L_R = torch.linalg.cholesky(c(R)).contiguous().requires_grad_(True) if sample else None
    packed = pack_mlp2(first, last)
    keep = [c(t) for t in (a, A_K, C_K, Q, R, m0, P0)]
    dims = capi.Dims(T, B, n, p, K, capi.KVAE_F64, 1, burn_in)
    fio = capi.fill(capi.FilterFwd(), a=keep[0], alpha=alpha, mask=mask, A_K=keep[1], C_K=keep[2],
                    Q=keep[3], R=keep[4], m_in=keep[5], P_in=keep[6], carry_per_sequence=0,
                    mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs,
                    status=status)
    lq, lr = (c(L_Q) if sample else None), (c(L_R) if sample else None)
    q = capi.fill(capi.MaskedMlp(), hidden=H, packed=packed, eps_z=eps_z, L_Q=lq, eps_a=eps_a,
                  L_R=lr, a_w=a_w)
    lib = hostsim_lib()
    fwd = lib.hostsim_filter_forward_masked_mlp
    fwd.argtypes = [C.POINTER(capi.Dims), C.POINTER(capi.FilterFwd), C.POINTER(capi.MaskedMlp)]
    assert fwd(C.byref(dims), C.byref(fio), C.byref(q)) == 0
    g_alpha = torch.cat([cot["weights"], new(1, B, K)]).contiguous()
    out = dict(d_a=new(T, B, p), d_alpha=new(T + 1, B, K), d_m_in=new(B, n), d_P_in=new(B, n, n),
               d_A_K=new(K, n, n), d_C_K=new(K, p, n), d_Q=new(n, n), d_R=new(p, p))
    bio = capi.FilterBwd()
    bio.fwd = fio
    capi.fill(bio, g_mean_filt=cot["filter_means"], g_cov_filt=cot["filter_covariances"],
              g_mean_pred=cot["filter_next_means"], g_cov_pred=cot["filter_next_covariances"],
              g_mat_As=cot["mat_As"], g_mat_Cs=cot["mat_Cs"], **out)
    d_logits, d_LQ, d_LR = new(T, B, K), new(n, n), new(p, p)
    nb = capi.MaskedMlpBwd()
    nb.fwd = q
    capi.fill(nb, g_a_w=cot["as_for_weight"], g_alpha=g_alpha, d_logits=d_logits,
              d_L_Q=d_LQ if sample else None, d_L_R=d_LR if sample else None,
              learn_weight_model=int(learn))
    bwd = lib.hostsim_filter_backward_masked_mlp
    bwd.argtypes = [C.POINTER(capi.Dims), C.POINTER(capi.FilterBwd), C.POINTER(capi.MaskedMlpBwd)]
    assert bwd(C.byref(dims), C.byref(bio), C.byref(nb)) == 0
    got = {"a": out["d_a"], "A_K": out["d_A_K"], "C_K": out["d_C_K"], "Q": out["d_Q"].clone(),
           "R": out["d_R"].clone(), "m0": out["d_m_in"].sum(0), "P0": out["d_P_in"].sum(0)}
    if sample:   # the imputation's factors are functions of Q and R
        got["Q"] += torch.autograd.grad(L_Q_of(Q), Q, d_LQ)[0]
        got["R"] += torch.autograd.grad(L_Q_of(R), R, d_LR)[0]
    # the network's parameters from d_logits, as the autograd node does
    x, dl = a_w.reshape(T * B, p), d_logits.reshape(T * B, K)
    pre = x @ first.weight.detach().T + first.bias.detach()
    dpre = (dl @ last.weight.detach()) * (pre > 0).to(dt)
    got.update(W1=dpre.T @ x, b1=dpre.sum(0), W2=dl.T @ torch.relu(pre), b2=dl.sum(0))
    for (name, val), ref in zip(got.items(), gref):
        if ref is None:
            assert float(val.abs().max()) == 0.0, name
            continue
        scale = float(ref.abs().max()) + 1e-30
        assert float((val - ref).abs().max()) / scale < 1e-9, (name, float((val - ref).abs().max()) / scale)


def L_Q_of(M):
    return torch.linalg.cholesky(M)
