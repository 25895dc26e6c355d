"""The two one-step adjoint kernels the masked LSTM training path is built from, each against
torch autograd of the same formula in float64:
kvae_lstm128_cell_backward (element-wise part of one LSTM step's adjoint, with the narrow-input
gradient) and kvae_impute_observation_backward (adjoint of the masked filter's imputation)."""
import ctypes as C

import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu
H = 128


@pytest.mark.parametrize("B,P,first,last", [(1, 2, False, False), (37, 1, False, True), (256, 2, True, False),
                                            (1000, 4, True, True), (16, 0, False, False)])
def test_cell_backward_matches_autograd(B, P, first, last):
    """first: no previous cell state (zero state); last: nothing arrives from a later step."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(B + P)
    dev = torch.device("cuda:0")
    pre = torch.randn(B, 4 * H, dtype=torch.float64, requires_grad=True)
    c_prev = None if first else torch.randn(B, H, dtype=torch.float64, requires_grad=True)
    i, f, g, o = pre.split(H, dim=1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c = i * g if first else f * c_prev + i * g
    h = o * torch.tanh(c)
    dh_a, dh_b, dc_in = (torch.randn(B, H, dtype=torch.float64) for _ in range(3))
    dh = dh_a if last else dh_a + dh_b
    loss = (h * dh).sum() if last else (h * dh).sum() + (c * dc_in).sum()
    wanted = torch.autograd.grad(loss, [pre] + ([] if first else [c_prev]))
    f32 = lambda t: None if t is None else t.detach().float().to(dev).contiguous()
    gates = f32(torch.cat([i, f, g, o], dim=1))
    dG, dcp = torch.empty(B, 4 * H, device=dev), torch.empty(B, H, device=dev)
    wih = torch.randn(4 * H, max(P, 1), device=dev)
    add = torch.randn(B, max(P, 1), device=dev)
    dx = torch.empty(B, max(P, 1), device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    ops = [f32(c_prev), f32(c), f32(dh_a), None if last else f32(dh_b), None if last else f32(dc_in)]
    rc = lib.kvae_lstm128_cell_backward(
        B, gates.data_ptr(), *(ptr(t) for t in ops), dG.data_ptr(), dcp.data_ptr(),
        wih.data_ptr() if P else None, P, add.data_ptr() if P else None, dx.data_ptr() if P else None,
        torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(dG.cpu().double(), wanted[0]) < 1e-5
    if not first:
        assert rel_err(dcp.cpu().double(), wanted[1]) < 1e-5
    if P:
        assert rel_err(dx, dG @ wih + add) < 1e-5


def test_cell_backward_checks_its_arguments():
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    dev = torch.device("cuda:0")
    t = lambda *s: torch.zeros(*s, device=dev)
    gates, c, dh, dG, dcp = t(4, 512), t(4, 128), t(4, 128), t(4, 512), t(4, 128)
    s = torch.cuda.current_stream().cuda_stream
    ok = lambda **k: lib.kvae_lstm128_cell_backward(
        k.get("B", 4), k.get("gates", gates.data_ptr()), None, c.data_ptr(), k.get("dh", dh.data_ptr()),
        None, None, dG.data_ptr(), dcp.data_ptr(), k.get("wih"), k.get("P", 0), None, k.get("dx"), s)
    assert ok() == 0
    assert ok(B=0) == -1 and ok(gates=None) == -1 and ok(dh=None) == -1
    assert ok(gates=gates.data_ptr() + 4) == -1                      # not 16-byte aligned
    assert ok(dx=t(4, 2).data_ptr(), P=2) == -1                      # a narrow-input gradient needs Wih
    assert ok(dx=t(4, 9).data_ptr(), P=9, wih=t(512, 9).data_ptr()) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("noise", [False, True])
@pytest.mark.parametrize("first", [False, True])
def test_impute_backward_matches_autograd(noise, first):
    """Every small (z_dim, a_dim) the library has kernels for.  first: the step has no previous
    step whose cotangents could take the gradient w.r.t. the predicted state."""
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd import functions as F
    lib = capi.load()
    dev = torch.device("cuda:0")
    done = set()
    for n, p, K in F.compiled_shapes(torch.float32):
        if (n, p) in done or n > 8 or not F.impute_supported(n, p, K, torch.float32, dev):
            continue
        done.add((n, p))
        torch.manual_seed(10 * n + p)
        B = 77
        d64 = lambda *s: torch.randn(*s, dtype=torch.float64)
        A_t, C_t, m_pred, a = (d64(B, n, n).requires_grad_(True), d64(B, p, n).requires_grad_(True),
                               d64(B, n).requires_grad_(True), d64(B, p).requires_grad_(True))
        mask = (torch.rand(B, dtype=torch.float64) < 0.5).double()
        mask[:5] = torch.rand(5, dtype=torch.float64)                 # fractional masks too
        eps_z, L_Q = (d64(B, n), torch.tril(d64(n, n)).requires_grad_(True)) if noise else (None, None)
        z = (A_t @ m_pred.unsqueeze(-1)).squeeze(-1)
        if noise:
            z = z + eps_z @ L_Q.T
        a_imp = (C_t @ z.unsqueeze(-1)).squeeze(-1)
        mu = mask.unsqueeze(-1)
        a_w = mu * a + (1 - mu) * a_imp
        g1, g2 = d64(B, p), d64(B, p)
        want = torch.autograd.grad((a_w * (g1 + g2)).sum(), [A_t, C_t, m_pred, a] + ([L_Q] if noise else []))
        f32 = lambda t: None if t is None else t.detach().float().to(dev).contiguous()
        d_a0, seg_m, seg_P, prev_m, prev_P = (f32(d64(B, p)), f32(d64(B, n)), f32(d64(B, n, n)),
                                              f32(d64(B, n)), f32(d64(B, n, n)))
        d_a, d_m_seg, g_mn, g_Pn = d_a0.clone(), seg_m.clone(), prev_m.clone(), prev_P.clone()
        d_aimp, dz = torch.empty(B, p, device=dev), torch.empty(B, n, device=dev)
        gC, gA = torch.empty(B, p, n, device=dev), torch.empty(B, n, n, device=dev)
        ptr = lambda t: None if t is None else t.data_ptr()
        dims = capi.Dims(1, B, n, p, K, capi.dtype_code(torch.float32), 1, 0)
        ops = [f32(g1), f32(g2), f32(mask), f32(A_t), f32(C_t), f32(m_pred), f32(eps_z), f32(L_Q)]
        rc = lib.kvae_impute_observation_backward(
            C.byref(dims), *(ptr(t) for t in ops), d_a.data_ptr(), d_aimp.data_ptr(), dz.data_ptr(),
            gC.data_ptr(), gA.data_ptr(), d_m_seg.data_ptr(), seg_P.data_ptr(),
            None if first else g_mn.data_ptr(), None if first else g_Pn.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
        assert rc == 0, (n, p)
        torch.cuda.synchronize()
        c = lambda t: t.cpu().double()
        tol = 2e-5
        assert rel_err(c(gA), want[0]) < tol and rel_err(c(gC), want[1]) < tol, (n, p)
        assert rel_err(c(d_a - d_a0), want[3]) < tol, (n, p)
        if first:
            assert rel_err(c(d_m_seg - seg_m), want[2]) < tol, (n, p)
        else:
            assert rel_err(c(g_mn - prev_m - seg_m), want[2]) < tol, (n, p)
            assert torch.equal(d_m_seg, seg_m)
            assert rel_err(c(g_Pn - prev_P), c(seg_P)) < tol, (n, p)
        if noise:
            assert rel_err((c(dz).unsqueeze(-1) * eps_z.unsqueeze(-2)).sum(0), want[4]) < tol, (n, p)
    assert len(done) >= 3
