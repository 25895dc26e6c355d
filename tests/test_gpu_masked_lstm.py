"""The masked filter with the default LSTM weight network in the recursion, under autograd
(masked_lstm.masked_filter_lstm_grad: the steps walked by hand in both directions through
kvae_lstm128_forward / kvae_lstm128_cell_backward / kvae_filter_backward and their companions)
against autograd through the float64 oracle's masked filter (oracle/kalman_oracle.py, pinned to
the reference by the golden fixtures) with the same torch.nn.LSTM and head in float64 on the
CPU: all eight outputs and the gradients of a random linear functional of them w.r.t.
observations, bases, noise covariances, initial state and every parameter of the network.
Tolerance (max-norm relative): 1e-4, the library's float32 bound; in both product forms."""
import pytest
import torch

from golden_util import rel_err
from oracle import kalman_oracle as O

pytestmark = pytest.mark.gpu

NAMES = ["filter_means", "filter_covariances", "filter_next_means", "filter_next_covariances",
         "mat_As", "mat_Cs", "as_for_weight", "weights"]
HID = 128


class SteppedNet:
    """softmax(linear(lstm(x))) called one step at a time, the state carried."""

    def __init__(self, lstm, linear):
        self.lstm, self.linear, self.state = lstm, linear, None

    def __call__(self, x):
        h, self.state = self.lstm(x, self.state)               # x (1,B,p)
        return torch.softmax(self.linear(h), dim=-1)


def run_case(n, p, K, sample, burn_in, learn, T=7, B=37, seed=0, exact=False, only=None):
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd import functions as F
    from kalman_vae_amd import masked_lstm
    torch.manual_seed(seed + n * 13 + p)
    dev = torch.device("cuda:0")
    dtype = torch.float32
    lstm, linear = torch.nn.LSTM(p, HID, num_layers=2), torch.nn.Linear(HID, K)
    A_K = 0.9 * torch.eye(n) + 0.3 / n * torch.randn(K, n, n)
    C_K = 0.5 * torch.randn(K, p, n)
    Lq = 0.4 * torch.eye(n) + 0.05 * torch.randn(n, n)
    Lr = 0.4 * torch.eye(p) + 0.05 * torch.randn(p, p)
    Q, R = Lq @ Lq.T + 0.01 * torch.eye(n), Lr @ Lr.T + 0.01 * torch.eye(p)
    Q, R = 0.5 * (Q + Q.T), 0.5 * (R + R.T)
    m0, P0 = 0.3 * torch.randn(n), 1.5 * torch.eye(n)
    a = torch.randn(T, B, p)
    mask = (torch.rand(T, B) < 0.6).to(dtype)
    mask[0] = 1.0
    eps_z = torch.randn(T, B, n) if sample else None
    eps_a = torch.randn(T, B, p) if sample else None

    f64 = lambda t: None if t is None else t.double()
    lstm64, lin64 = torch.nn.LSTM(p, HID, num_layers=2).double(), torch.nn.Linear(HID, K).double()
    lstm64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    lin64.load_state_dict({k: v.double() for k, v in linear.state_dict().items()})
    leaves = [f64(t).requires_grad_(True) for t in (a, A_K, C_K, Q, R, m0, P0)]
    want = O.kalman_filter(*leaves, SteppedNet(lstm64, lin64), mask=f64(mask), eps_state=f64(eps_z),
                           eps_obs=f64(eps_a), learn_weight_model=learn, burn_in=burn_in)
    # (only: the functional looks at these outputs alone; the others get no cotangent at all)
    cot = {k: torch.randn_like(want[k]) if only is None or k in only else None for k in NAMES}
    loss = sum((cot[k] * want[k]).sum() for k in NAMES if cot[k] is not None)
    net_names = [k for k, _ in lstm64.named_parameters()] + ["head.weight", "head.bias"]
    params = leaves + list(lstm64.parameters()) + [lin64.weight, lin64.bias]
    gref = torch.autograd.grad(loss, params, allow_unused=True)

    d = lambda t: None if t is None else t.to(dev)
    lstm, linear = lstm.to(dev), linear.to(dev)
    gl = [d(t).requires_grad_(True) for t in (a, A_K, C_K, Q, R, m0, P0)]
    L_Q = F.cholesky_factor(gl[3]) if sample else None
    L_R = F.cholesky_factor(gl[4]) if sample else None
    lib = capi.load()
    before = lib.kvae_lstm128_get_precision()
    lib.kvae_lstm128_set_precision(0 if exact else 1)
    try:
        out = masked_lstm.masked_filter_lstm_grad(
            gl[0], d(mask), lstm, linear, gl[1], gl[2], gl[3], gl[4], gl[5], gl[6], d(eps_z), L_Q,
            d(eps_a), L_R, burn_in=burn_in, learn_weight_model=learn)
        assert out is not None
        got = dict(zip(NAMES, out[:8]))
        got["weights"] = got["weights"][:T]
        loss_g = sum((d(cot[k]).to(dtype) * got[k]).sum() for k in NAMES if cot[k] is not None)
        gp = gl + list(lstm.parameters()) + [linear.weight, linear.bias]
        ggot = torch.autograd.grad(loss_g, gp, allow_unused=True)
    finally:
        lib.kvae_lstm128_set_precision(before)
    tol = 1e-4
    bad = []
    for k in NAMES:
        e = rel_err(got[k], want[k])
        if not e < tol:
            bad.append(f"{k}: {e:.2e}")
    pn = ["a", "A_K", "C_K", "Q", "R", "m0", "P0"] + net_names
    for name, g, r in zip(pn, ggot, gref):
        if r is None or float(r.abs().max()) == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, name
            continue
        assert g is not None, name
        e = rel_err(g, r)
        if not e < tol:
            bad.append(f"grad {name}: {e:.2e}")
    h_n, c_n = out[8], out[9]
    assert h_n.shape == (2, B, HID) and c_n.shape == (2, B, HID)
    assert F.read_status(dev) == 0
    assert not bad, "; ".join(bad)


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("sample,burn_in,learn", [(False, 0, True), (True, 2, True),
                                                  (True, 0, False)])
@pytest.mark.parametrize("n,p,K", [(4, 2, 3), (2, 1, 2)])
def test_masked_lstm_filter_matches_oracle_autograd(n, p, K, sample, burn_in, learn, exact):
    run_case(n, p, K, sample, burn_in, learn, exact=exact)


@pytest.mark.parametrize("learn", [True, False])
@pytest.mark.parametrize("only", [("filter_means",), ("filter_next_covariances", "mat_Cs"),
                                  ("as_for_weight",), ("weights",)])
def test_functional_of_some_outputs_only(only, learn):
    """Outputs without a cotangent (autograd hands the node None for them), among them the case
    in which nothing reaches the imputed observation: detached weights and no cotangent of a_w."""
    if only == ("weights",) and not learn:
        pytest.skip("detached weights: nothing to differentiate")
    run_case(4, 2, 3, True, 1, learn, only=only)


def test_one_step_sequence():
    run_case(4, 2, 3, True, 0, True, T=1, B=16)


def test_module_masked_training_step_with_the_lstm_network_is_one_node(monkeypatch):
    """StateSpaceModel.kalman_filter with a mask and the default LSTM network under autograd
    goes through the single-node path (no per-step autograd nodes, no library LSTM call), leaves
    the network's state as the per-step calls would, and counts its calls."""
    from kalman_vae_amd import SampleControl, StateSpaceModel
    from kalman_vae_amd import functions as F
    from kalman_vae_amd import masked_lstm
    calls, steps = [], []
    real = masked_lstm.masked_filter_lstm_grad
    monkeypatch.setattr(masked_lstm, "masked_filter_lstm_grad",
                        lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    real_steps = F.kalman_filter_steps
    monkeypatch.setattr(F, "kalman_filter_steps",
                        lambda *a, **k: (steps.append(1), real_steps(*a, **k))[1])
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ssm = StateSpaceModel(2, 4, 3, "lstm", hidden_dim=128).to(dev).train()
    a = torch.randn(10, 33, 2, device=dev, requires_grad=True)
    mask = (torch.rand(10, 33, device=dev) < 0.7).float()
    torch.manual_seed(1)
    out = ssm.kalman_filter(a, SampleControl(), observation_mask=mask)
    (out[0].sum() + out[6].sum()).backward()
    assert calls == [1] and steps == []
    assert a.grad is not None and torch.isfinite(a.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all()
               for p in ssm.weight_model.parameters())
    h_n, c_n = ssm.weight_model.hidden
    assert h_n.shape == (2, 33, 128) and ssm.weight_model.count == 10
    # the same call through the step-by-step path (library LSTM, per-step autograd nodes)
    # (in eval mode: the module then takes PyTorch's native fp32 LSTM kernels, not MIOpen's)
    ref = StateSpaceModel(2, 4, 3, "lstm", hidden_dim=128).to(dev).eval()
    ref.load_state_dict(ssm.state_dict())
    ref.weight_model.use_hip_recurrence = False
    a2 = a.detach().clone().requires_grad_(True)
    torch.manual_seed(1)
    out2 = ref.kalman_filter(a2, SampleControl(), observation_mask=mask)
    (out2[0].sum() + out2[6].sum()).backward()
    assert len(steps) == 10
    for x, y in zip(out, out2):
        assert rel_err(x, y) < 1e-4
    assert rel_err(a.grad, a2.grad) < 1e-4
    for (k, p), q in zip(ssm.named_parameters(), ref.parameters()):
        if q.grad is not None and float(q.grad.abs().max()) > 0:
            assert rel_err(p.grad, q.grad) < 2e-4, k


@pytest.mark.parametrize("name", ["masked_lstm128_f64", "masked_lstm128_mean_f64"])
def test_reference_fixture_goes_through_the_node(name, monkeypatch):
    """The reference's own run of a masked sequence with its default 2 x 128 LSTM (float64
    fixture: outputs and gradients of every parameter) replayed in float32: the module takes
    the one-node path, and meets the reference at the library's float32 bound."""
    from golden_util import Golden
    from kalman_vae_amd import masked_lstm
    from test_gpu_module_golden import compare, run_module
    calls = []
    real = masked_lstm.masked_filter_lstm_grad
    monkeypatch.setattr(masked_lstm, "masked_filter_lstm_grad",
                        lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    g = Golden(name)
    obj, info, grads = run_module(g, torch.float32, monkeypatch)
    assert calls == [1]
    compare(g, obj, info, grads, 1e-4)


def test_masked_step_with_the_lstm_network_replays_as_a_graph():
    """The whole masked ELBO step with the default LSTM network (the one-node filter, smoother,
    ELBO terms, forward + backward) captured as a HIP graph: no memset node (the node's
    reductions are the library's), and the replay gives the objective and the gradients of the
    eager step from the same generator state."""
    from kalman_vae_amd import SampleControl, StateSpaceModel, kalman_elbo
    from kalman_vae_amd.graphed import GraphedElboStep
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    ssm = StateSpaceModel(a_dim=2, z_dim=4, K=3, dynamics_parameter_network="lstm").to(dev).train()
    ssm.rng_parity = False
    sc = SampleControl()
    mask = (torch.rand(20, 96, device=dev) < 0.7).float()
    a1 = torch.randn(20, 96, 2, device=dev) * 0.3
    a2 = torch.randn(20, 96, 2, device=dev) * 0.3
    step = GraphedElboStep(ssm, a1, sc, burn_in=3, observation_mask=mask)
    assert step.kernel_nodes > 100
    for a in (a1, a2):
        state = torch.cuda.get_rng_state(dev)
        obj_g = step(a).clone()
        grads_g = {n: p.grad.clone() for n, p in ssm.named_parameters() if p.grad is not None}
        a_grad_g = step.as_grad.clone()
        torch.cuda.set_rng_state(state, dev)
        ssm.zero_grad(set_to_none=True)
        a_e = a.clone().requires_grad_(True)
        obj_e, _ = kalman_elbo(ssm, a_e, sc, burn_in=3, observation_mask=mask)
        (-obj_e).backward()
        grads_e = {n: p.grad for n, p in ssm.named_parameters() if p.grad is not None}
        assert set(grads_g) == set(grads_e) and grads_e
        if rel_err(obj_g, obj_e.detach()) > 1e-5:
            pytest.skip("graph replay draws different noise than the eager step on this torch build")
        assert rel_err(a_grad_g, a_e.grad) < 1e-4
        bad = {n: rel_err(grads_g[n], e) for n, e in grads_e.items()
               if not rel_err(grads_g[n], e) < 1e-4}
        assert not bad, bad
