"""Kernel-vs-oracle comparisons shared by the CPU (host build of the device
source) and GPU (HIP library) test modules.  Each check runs one entry point
forward and backward on a seeded random problem and compares every output and
gradient with oracle/kalman_oracle.py evaluated in float64.

Tolerances: float64 kernels 1e-9 relative (max-norm, per tensor); float32
kernels 1e-4 relative for outputs and gradients, the bar the fp32 target is
held to.  The oracle always runs in float64 on the float32-rounded inputs."""
import torch

from golden_util import rel_err
from kernel_harness import AlphaReplay, random_problem
from oracle import kalman_oracle as O

TOL = {torch.float64: 1e-9, torch.float32: 1e-4}


def _dev(backend, prob):
    return {k: (v.to(backend.device).contiguous() if v is not None else None)
            for k, v in prob.items()}


def _f64(prob, names, grad=True):
    out = {}
    for k in names:
        v = prob[k]
        out[k] = None if v is None else v.double().clone().requires_grad_(grad)
    return out


def _cmp(name, got, want, tol, failures):
    e = rel_err(got, want)
    if not (e < tol):
        failures.append(f"{name}: rel err {e:.3e} (tol {tol:.1e})")


def _cots(tensors, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(v.shape, generator=g, dtype=torch.float64) for k, v in tensors.items()}


def check_filter(backend, T, B, n, p, K, dtype, seed=0, mask_kind="none", symmetrize=True,
                 burn_in=0, per_sequence_carry=False):
    tol = TOL[dtype]
    prob = random_problem(T, B, n, p, K, dtype, seed, mask_kind)
    if per_sequence_carry:
        g = torch.Generator().manual_seed(seed + 77)
        m_in = (0.3 * torch.randn(B, n, generator=g, dtype=torch.float64)).to(dtype)
        W = torch.randn(B, n, n, generator=g, dtype=torch.float64)
        P_in = (0.5 * torch.eye(n, dtype=torch.float64) + 0.1 * W @ W.transpose(1, 2)).to(dtype)
    else:
        m_in, P_in = prob["m0"], prob["P0"]
    dv = _dev(backend, dict(prob, m_in=m_in, P_in=P_in))
    fwd = backend.filter_forward(dv["a"], dv["alpha"], dv["mask"], dv["A_K"], dv["C_K"], dv["Q"],
                                 dv["R"], dv["m_in"], dv["P_in"], symmetrize=symmetrize)
    assert int(fwd["status"].item()) == 0

    # oracle in float64
    o = _f64(dict(prob, m_in=m_in, P_in=P_in), ["a", "alpha", "A_K", "C_K", "Q", "R", "m_in", "P_in"])
    mask64 = None if prob["mask"] is None else prob["mask"].double()
    # a per-sequence carry is (B,n)/(B,n,n); the oracle's expand() is then a no-op
    ref = O.kalman_filter(o["a"], o["A_K"], o["C_K"], o["Q"], o["R"], o["m_in"], o["P_in"],
                          AlphaReplay(o["alpha"]), mask=mask64, symmetrize=symmetrize,
                          burn_in=burn_in, alpha0=o["alpha"][0])
    names = {"mean_filt": "filter_means", "cov_filt": "filter_covariances",
             "mean_pred": "filter_next_means", "cov_pred": "filter_next_covariances",
             "mat_As": "mat_As", "mat_Cs": "mat_Cs"}
    failures = []
    for k, rk in names.items():
        _cmp("fwd." + k, fwd[k], ref[rk], tol, failures)

    cots = _cots({k: ref[rk] for k, rk in names.items()}, seed + 1)
    loss = sum((cots[k] * ref[rk]).sum() for k, rk in names.items())
    wrt = ["a", "alpha", "A_K", "C_K", "Q", "R", "m_in", "P_in"]
    gref = torch.autograd.grad(loss, [o[k] for k in wrt], allow_unused=True)
    gref = {k: (torch.zeros_like(o[k]) if g is None else g) for k, g in zip(wrt, gref)}
    grads = {k: cots[k].to(dtype).to(backend.device).contiguous() for k in names}
    bwd = backend.filter_backward(fwd, grads, symmetrize=symmetrize, burn_in=burn_in)
    _cmp("d_a", bwd["d_a"], gref["a"], tol, failures)
    _cmp("d_alpha", bwd["d_alpha"], gref["alpha"], tol, failures)
    _cmp("d_A_K", bwd["d_A_K"], gref["A_K"], tol, failures)
    _cmp("d_C_K", bwd["d_C_K"], gref["C_K"], tol, failures)
    _cmp("d_Q", bwd["d_Q"], gref["Q"], tol, failures)
    _cmp("d_R", bwd["d_R"], gref["R"], tol, failures)
    if per_sequence_carry:
        _cmp("d_m_in", bwd["d_m_in"], gref["m_in"], tol, failures)
        _cmp("d_P_in", bwd["d_P_in"], gref["P_in"], tol, failures)
    else:
        _cmp("d_m_in", bwd["d_m_in"].sum(0), gref["m_in"], tol, failures)
        _cmp("d_P_in", bwd["d_P_in"].sum(0), gref["P_in"], tol, failures)
    assert not failures, "\n".join(failures)
    return fwd, ref


def check_smooth(backend, T, B, n, p, K, dtype, seed=0, sample=True, symmetrize=True, burn_in=0):
    tol = TOL[dtype]
    prob = random_problem(T, B, n, p, K, dtype, seed)
    prob["alpha"][0] = 1.0 / K
    # inputs of the smoother: a float64 oracle filter pass, rounded to dtype
    f = O.kalman_filter(prob["a"].double(), prob["A_K"].double(), prob["C_K"].double(),
                        prob["Q"].double(), prob["R"].double(), prob["m0"].double(),
                        prob["P0"].double(), AlphaReplay(prob["alpha"].double()),
                        symmetrize=symmetrize)
    ins = {"m": f["filter_means"], "P": f["filter_covariances"], "mn": f["filter_next_means"],
           "Pn": f["filter_next_covariances"], "As": f["mat_As"], "Cs": f["mat_Cs"]}
    ins = {k: v.detach().to(dtype) for k, v in ins.items()}
    LR = torch.linalg.cholesky(prob["R"].double()).to(dtype)
    es = prob["eps_s"] if sample else None
    eo = prob["eps_o"] if sample else None
    dv = _dev(backend, dict(ins, LR=LR, es=es, eo=eo))
    fwd = backend.smooth_forward(dv["m"], dv["P"], dv["mn"], dv["Pn"], dv["As"], dv["Cs"],
                                 LR=dv["LR"], eps_state=dv["es"], eps_obs=dv["eo"],
                                 symmetrize=symmetrize)
    assert int(fwd["status"].item()) == 0

    o = {k: v.double().clone().requires_grad_(True) for k, v in ins.items()}
    # differentiate w.r.t. LR through R = LR LR^T (chol(R) == LR for a valid factor)
    LRo = LR.double().clone().requires_grad_(True)

    ref = _oracle_smooth_with_LR(o, LRo, None if es is None else es.double(),
                                 None if eo is None else eo.double(), symmetrize, burn_in)
    names = {"mean_smooth": "means", "cov_smooth": "covariances", "zs": "zs",
             "as_resampled": "as_resampled"}
    failures = []
    for k, rk in names.items():
        _cmp("fwd." + k, fwd[k], ref[rk], tol, failures)
    cots = _cots({k: ref[rk] for k, rk in names.items()}, seed + 2)
    loss = sum((cots[k] * ref[rk]).sum() for k, rk in names.items())
    wrt = list(o.values()) + [LRo]
    gref = torch.autograd.grad(loss, wrt, allow_unused=True)
    gref = [torch.zeros_like(w) if g is None else g for w, g in zip(wrt, gref)]
    gref = dict(zip(list(o.keys()) + ["LR"], gref))
    grads = {k: cots[k].to(dtype).to(backend.device).contiguous() for k in names}
    bwd = backend.smooth_backward(fwd, grads, symmetrize=symmetrize, burn_in=burn_in)
    for k, gk in (("d_mean_filt", "m"), ("d_cov_filt", "P"), ("d_mean_pred", "mn"),
                  ("d_cov_pred", "Pn"), ("d_mat_As", "As"), ("d_mat_Cs", "Cs")):
        _cmp(k, bwd[k], gref[gk], tol, failures)
    if sample:
        _cmp("d_LR", bwd["d_LR"], gref["LR"], tol, failures)
    assert not failures, "\n".join(failures)


def _oracle_smooth_with_LR(o, LR, es, eo, symmetrize, burn_in):
    """oracle smoother with the observation-noise factor given directly, so
    that the gradient w.r.t. the factor itself can be compared."""
    R = LR @ LR.T
    # O.kalman_smooth recomputes chol(R); its derivative w.r.t. LR differs from
    # the kernel's d_LR (which treats LR as the input).  Re-emit by hand.
    s = O.kalman_smooth(o["m"], o["P"], o["mn"], o["Pn"], o["As"], o["Cs"], R.detach(),
                        eps_state=es, eps_obs=None, symmetrize=symmetrize, burn_in=burn_in)
    if eo is not None:
        T = o["m"].shape[0]
        Cs = torch.cat([o["Cs"][:T - 1], o["Cs"][T:]])  # C_0..C_{T-2}, C_T
        s["as_resampled"] = torch.matmul(Cs, s["zs"].unsqueeze(-1)).squeeze(-1) + eo @ LR.T
    return s


def check_elbo(backend, T, B, n, p, K, dtype, seed=0):
    tol = TOL[dtype]
    prob = random_problem(T, B, n, p, K, dtype, seed)
    prob["alpha"][0] = 1.0 / K
    with torch.no_grad():
        f = O.kalman_filter(prob["a"].double(), prob["A_K"].double(), prob["C_K"].double(),
                            prob["Q"].double(), prob["R"].double(), prob["m0"].double(),
                            prob["P0"].double(), AlphaReplay(prob["alpha"].double()))
        s = O.kalman_smooth(f["filter_means"], f["filter_covariances"], f["filter_next_means"],
                            f["filter_next_covariances"], f["mat_As"], f["mat_Cs"],
                            prob["R"].double())
    ins = {"a": prob["a"], "ms": s["means"], "Ps": s["covariances"], "As": f["mat_As"],
           "Cs": f["mat_Cs"], "Q": prob["Q"], "R": prob["R"], "m0": prob["m0"], "P0": prob["P0"]}
    ins = {k: v.detach().to(dtype) for k, v in ins.items()}
    eps = prob["eps_e"]
    dv = _dev(backend, dict(ins, eps=eps))
    fwd = backend.elbo_forward(dv["a"], dv["ms"], dv["Ps"], dv["As"], dv["Cs"], dv["Q"], dv["R"],
                               dv["m0"], dv["P0"], dv["eps"])
    assert int(fwd["status"].item()) == 0
    o = {k: v.double().clone().requires_grad_(True) for k, v in ins.items()}
    ref = O.elbo_terms(o["a"], o["ms"], o["Ps"], o["As"], o["Cs"], o["Q"], o["R"], o["m0"],
                       o["P0"], eps.double())
    failures = []
    for k in ("obs", "trans", "post", "z"):
        _cmp("fwd." + k, fwd[k], ref[k], tol, failures)
    cots = _cots({k: ref[k] for k in ("obs", "trans", "post")}, seed + 3)
    loss = sum((cots[k] * ref[k]).sum() for k in cots)
    gref = torch.autograd.grad(loss, list(o.values()), allow_unused=True)
    gref = {k: (torch.zeros_like(o[k]) if g is None else g) for k, g in zip(o.keys(), gref)}
    gd = {k: cots[k].to(dtype).to(backend.device).contiguous() for k in cots}
    bwd = backend.elbo_backward(fwd, gd["obs"], gd["trans"], gd["post"])
    for k, gk in (("d_a", "a"), ("d_mean_smooth", "ms"), ("d_cov_smooth", "Ps"),
                  ("d_mat_As", "As"), ("d_mat_Cs", "Cs"), ("d_Q", "Q"), ("d_R", "R"),
                  ("d_m0", "m0"), ("d_P0", "P0")):
        _cmp(k, bwd[k], gref[gk], tol, failures)
    assert not failures, "\n".join(failures)


def check_loss(backend, T, B, n, p, K, dtype, seed=0, mask_kind="none", symmetrize=True,
               burn_in=0, weights=(0.7, 1.3, -0.9), stride=1):
    """kvae_kalman_loss_forward / _backward (filter -> smoother -> ELBO terms -> weighted
    totals in one call per direction) against the oracle's three separate stages and its
    autograd: per-pair terms, the four outputs, d_a, d_alpha and every parameter gradient."""
    tol = TOL[dtype]
    prob = random_problem(T, B, n, p, K, dtype, seed, mask_kind)
    dv = _dev(backend, prob)
    fwd = backend.loss_forward(dv["a"], dv["alpha"], dv["mask"], dv["A_K"], dv["C_K"], dv["Q"],
                               dv["R"], dv["m0"], dv["P0"], dv["eps_e"], weights,
                               symmetrize=symmetrize)
    assert int(fwd["status"].item()) == 0

    wrt = ["a", "alpha", "A_K", "C_K", "Q", "R", "m0", "P0"]
    o = _f64(prob, wrt)
    mask64 = None if prob["mask"] is None else prob["mask"].double()
    f = O.kalman_filter(o["a"], o["A_K"], o["C_K"], o["Q"], o["R"], o["m0"], o["P0"],
                        AlphaReplay(o["alpha"]), mask=mask64, symmetrize=symmetrize,
                        burn_in=burn_in, alpha0=o["alpha"][0])
    s = O.kalman_smooth(f["filter_means"], f["filter_covariances"], f["filter_next_means"],
                        f["filter_next_covariances"], f["mat_As"], f["mat_Cs"], o["R"],
                        symmetrize=symmetrize, burn_in=burn_in)
    e = O.elbo_terms(o["a"], s["means"], s["covariances"], f["mat_As"], f["mat_Cs"], o["Q"],
                     o["R"], o["m0"], o["P0"], prob["eps_e"].double())
    totals = torch.stack([w * e[k].sum() for w, k in zip(weights, ("obs", "trans", "post"))])
    failures = []
    for i, k in enumerate(("obs", "trans", "post")):
        _cmp("terms." + k, fwd["terms"][i], e[k], tol, failures)
    # the totals are sums of T*B terms of mixed sign: compare on the scale of the terms
    scale = sum(abs(w) * e[k].abs().sum() for w, k in zip(weights, ("obs", "trans", "post")))
    got = fwd["out"].double().cpu()
    if not (float((got[:3] - totals.detach()).abs().max()) < tol * float(scale.detach())):
        failures.append(f"out[0:3]: {got[:3].tolist()} vs {totals.tolist()}")
    if not (abs(float(got[3] - totals.detach().sum())) < tol * float(scale.detach())):
        failures.append(f"out[3]: {float(got[3])} vs {float(totals.detach().sum())}")

    # cotangents of the three weighted totals (stride 1) or of their total (stride 0)
    cot = torch.tensor([0.8, -1.1, 0.6], dtype=torch.float64) if stride else \
        torch.tensor([1.7], dtype=torch.float64)
    loss = (cot * totals).sum() if stride else cot[0] * totals.sum()
    gref = torch.autograd.grad(loss, [o[k] for k in wrt], allow_unused=True)
    gref = {k: (torch.zeros_like(o[k]) if g is None else g) for k, g in zip(wrt, gref)}
    g_scale = cot.to(dtype).to(backend.device).contiguous()
    bwd = backend.loss_backward(fwd, g_scale, stride, symmetrize=symmetrize, burn_in=burn_in)
    for k in wrt:
        _cmp("d_" + k, bwd["d_" + k], gref[k], tol, failures)
    if backend.kind == "hip":
        # the same results from the one-call form
        both = backend.loss_value_and_grad(dv["a"], dv["alpha"], dv["mask"], dv["A_K"], dv["C_K"],
                                           dv["Q"], dv["R"], dv["m0"], dv["P0"], dv["eps_e"],
                                           weights, g_scale, stride, symmetrize=symmetrize,
                                           burn_in=burn_in)
        assert int(both["status"].item()) == 0
        for i, k in enumerate(("obs", "trans", "post")):
            _cmp("both.terms." + k, both["terms"][i], e[k], tol, failures)
        got = both["out"].double().cpu()
        if not (float((got[:3] - totals.detach()).abs().max()) < tol * float(scale.detach())):
            failures.append(f"both.out[0:3]: {got[:3].tolist()} vs {totals.tolist()}")
        if not (abs(float(got[3] - totals.detach().sum())) < tol * float(scale.detach())):
            failures.append(f"both.out[3]: {float(got[3])} vs {float(totals.detach().sum())}")
        for k in wrt:
            _cmp("both.d_" + k, both["d_" + k], gref[k], tol, failures)
    assert not failures, "\n".join(failures)


def check_masked_mlp(device, T, B, n, p, K, hidden, dtype, seed=0, mask_kind="binary",
                     sample=False, symmetrize=True):
    """kvae_filter_forward_masked_mlp (HIP library, through functions.masked_filter_mlp)
    against the oracle's masked filter whose weight function is the same perceptron."""
    from kalman_vae_amd import functions as F
    tol = TOL[dtype]
    prob = random_problem(T, B, n, p, K, dtype, seed, mask_kind)
    torch.manual_seed(seed)
    first, last = torch.nn.Linear(p, hidden).to(dtype), torch.nn.Linear(hidden, K).to(dtype)
    eps_z = prob["eps_s"] if sample else None
    eps_a = prob["eps_o"] if sample else None
    f64 = lambda t: None if t is None else t.double()
    f1, l1 = first.double(), last.double()      # float64 copies of the rounded parameters
    with torch.no_grad():
        net = lambda x: torch.softmax(l1(torch.relu(f1(x))), dim=-1)
        want = O.kalman_filter(f64(prob["a"]), f64(prob["A_K"]), f64(prob["C_K"]), f64(prob["Q"]),
                               f64(prob["R"]), f64(prob["m0"]), f64(prob["P0"]), net,
                               mask=f64(prob["mask"]), eps_state=f64(eps_z), eps_obs=f64(eps_a),
                               symmetrize=symmetrize)
        L_Q = torch.linalg.cholesky(f64(prob["Q"])).to(dtype) if sample else None
        L_R = torch.linalg.cholesky(f64(prob["R"])).to(dtype) if sample else None
    d = lambda t: None if t is None else t.to(device).contiguous()
    packed = F.pack_mlp2(f1, l1).to(dtype)
    got = F.masked_filter_mlp(d(prob["a"]), d(prob["mask"]), d(packed), hidden, d(prob["A_K"]),
                              d(prob["C_K"]), d(prob["Q"]), d(prob["R"]), d(prob["m0"]),
                              d(prob["P0"]), d(eps_z), d(L_Q), d(eps_a), d(L_R),
                              symmetrize=symmetrize)
    assert got is not None, "no kernel for these dimensions"
    assert F.read_status(torch.device(device)) == 0
    names = ["filter_means", "filter_covariances", "filter_next_means", "filter_next_covariances",
             "mat_As", "mat_Cs", "as_for_weight", "weights"]
    failures = []
    for name, val in zip(names, got):
        val = val[:T] if name == "weights" else val
        _cmp(name, val.cpu(), want[name], tol, failures)
    assert not failures, "; ".join(failures)
