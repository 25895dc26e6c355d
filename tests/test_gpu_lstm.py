"""HIP LSTM recurrence against torch.nn.LSTM (PyTorch-native kernels): outputs,
final states and gradients w.r.t. input and all parameters, float32.
Tolerance 1e-4 relative (max-norm) -- both sides are fp32 with different
summation orders."""
import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture
def exact_products():
    """The resident kernels with the exact fp32 matrix instruction for the test's duration."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    assert lib.kvae_lstm128_set_precision(0) == 0 and lib.kvae_lstm128_get_precision() == 0
    yield
    assert lib.kvae_lstm128_set_precision(1) == 0


@pytest.mark.parametrize("persistent", ["resident", "resident-exact", 16, 32, 0])
@pytest.mark.parametrize("T,B,p,H,L", [(20, 70, 2, 128, 2), (5, 33, 3, 32, 1), (7, 64, 2, 64, 3),
                                       (1, 40, 2, 128, 2), (3, 1, 16, 128, 2), (9, 32, 4, 128, 2),
                                       (4, 200, 1, 128, 2)])
def test_fused_lstm_matches_torch(T, B, p, H, L, persistent, monkeypatch, request):
    import kalman_vae_amd.lstm_function as lf
    from kalman_vae_amd.lstm_function import fused_lstm
    # persistent: tile rows of the one-launch two-layer kernels (both forms are
    # forced in turn; the library picks by batch size otherwise), 0 = per-step kernels
    # "resident": one launch per layer, recurrent matrix in registers (the default: products
    # as three bf16 matrix instructions, with the row-streaming and weight-gradient kernels
    # around them); "resident-exact": the same with the fp32 matrix instruction and library
    # GEMMs around them
    if persistent == "resident-exact":
        request.getfixturevalue("exact_products")
        persistent = "resident"
    monkeypatch.setattr(lf, "PERSISTENT", bool(persistent))
    monkeypatch.setattr(lf, "RESIDENT", persistent == "resident")
    if persistent and persistent != "resident":
        monkeypatch.setenv("KVAE_LSTM_TILE_ROWS", str(persistent))
    torch.manual_seed(T * 100 + B)
    dev = torch.device("cuda:0")
    lstm = torch.nn.LSTM(p, H, num_layers=L).to(dev)
    x = torch.randn(T, B, p, device=dev, requires_grad=True)
    w = torch.randn(T, B, H, device=dev)
    with torch.backends.cudnn.flags(enabled=False):
        ref, (hn, cn) = lstm(x)
        gref = torch.autograd.grad((ref * w).sum(), [x] + list(lstm.parameters()))
    out, (hn2, cn2) = fused_lstm(x, lstm)
    g = torch.autograd.grad((out * w).sum(), [x] + list(lstm.parameters()))
    assert rel_err(out, ref) < 1e-5
    assert rel_err(hn2, hn) < 1e-5 and rel_err(cn2, cn) < 1e-5
    names = ["x"] + [n for n, _ in lstm.named_parameters()]
    bad = [(n, rel_err(a, b)) for n, a, b in zip(names, g, gref) if not rel_err(a, b) < 1e-4]
    assert not bad, bad


def test_weight_model_whole_sequence_equals_stepwise():
    """LSTMModel: one fused whole-sequence call == the reference's step-by-step
    calls through torch.nn.LSTM, including the cached final state."""
    from kalman_vae_amd.dynamics_parameter_network import LSTMModel
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    net = LSTMModel(2, 3, 128, 2).to(dev)
    x = torch.randn(12, 50, 2, device=dev)
    net.clear_hidden_state()
    whole = net(x)
    h_whole = net.hidden
    net.clear_hidden_state()
    with torch.backends.cudnn.flags(enabled=False):
        steps = torch.cat([net(x[t:t + 1]) for t in range(12)])
    assert rel_err(whole, steps) < 1e-5
    assert rel_err(h_whole[0], net.hidden[0]) < 1e-5
    assert rel_err(h_whole[1], net.hidden[1]) < 1e-5


@pytest.mark.parametrize("R_shape,H,K", [((20, 70), 128, 3), ((7, 33), 32, 2), ((1000,), 64, 8),
                                         ((3, 5, 11), 256, 1), ((20, 4096), 128, 3), ((17,), 128, 5)])
def test_softmax_head_matches_torch(R_shape, H, K):
    """Linear+softmax head kernels: output, d_h, dW, db against torch (fp32 both
    sides, different summation order: 1e-4 relative in max-norm)."""
    from kalman_vae_amd.lstm_function import softmax_head
    torch.manual_seed(H + K)
    dev = torch.device("cuda:0")
    lin = torch.nn.Linear(H, K).to(dev)
    h = torch.randn(*R_shape, H, device=dev, requires_grad=True)
    w = torch.randn(*R_shape, K, device=dev)
    ref = torch.softmax(lin(h), dim=-1)
    gref = torch.autograd.grad((ref * w).sum(), [h, lin.weight, lin.bias])
    out = softmax_head(h, lin)
    g = torch.autograd.grad((out * w).sum(), [h, lin.weight, lin.bias])
    assert rel_err(out, ref) < 1e-5
    bad = [(n, rel_err(a, b)) for n, a, b in zip(("h", "W", "b"), g, gref) if not rel_err(a, b) < 1e-4]
    assert not bad, bad


@pytest.mark.parametrize("B,rows", [(50, 16), (50, 32), (1, 16), (300, 32)])
def test_weight_model_stepwise_inference_on_hip_equals_torch(B, rows, monkeypatch):
    """Step-by-step inference (the masked filter's call pattern) through the
    persistent kernel with carried state == torch.nn.LSTM fed the same steps."""
    from kalman_vae_amd.dynamics_parameter_network import LSTMModel
    monkeypatch.setenv("KVAE_LSTM_TILE_ROWS", str(rows))
    torch.manual_seed(B)
    dev = torch.device("cuda:0")
    net = LSTMModel(2, 3, 128, 2).to(dev).eval()
    x = torch.randn(9, B, 2, device=dev)
    with torch.no_grad():
        net.clear_hidden_state()
        hip = torch.cat([net(x[t:t + 1]) for t in range(5)] + [net(x[5:9])])   # 1-step and 4-step calls
        h_hip = net.hidden
        net.clear_hidden_state()
        net.use_hip_recurrence = False
        with torch.backends.cudnn.flags(enabled=False):
            ref = torch.cat([net(x[t:t + 1]) for t in range(9)])
        h_ref = net.hidden
    assert rel_err(hip, ref) < 1e-5
    assert rel_err(h_hip[0], h_ref[0]) < 1e-5 and rel_err(h_hip[1], h_ref[1]) < 1e-5


@pytest.mark.parametrize("T,B,P", [(6, 37, 0), (4, 16, 2), (1, 5, 3), (3, 100, 4)])
def test_resident_layer_with_carried_state_matches_torch(T, B, P):
    """kvae_lstm128_forward / _backward through the C ABI on ONE layer that starts from a
    given (h, c): outputs, dG-derived gradients, dx and the bias / input-weight sums
    against torch.nn.LSTM (fp32 both sides)."""
    import ctypes as C
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(T * 10 + B)
    dev = torch.device("cuda:0")
    H = 128
    in_dim = P if P > 0 else H
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev)
    x = torch.randn(T, B, in_dim, device=dev, requires_grad=True)
    h_in = (0.5 * torch.randn(1, B, H, device=dev)).requires_grad_(False)
    c_in = 0.5 * torch.randn(1, B, H, device=dev)
    wgt = torch.randn(T, B, H, device=dev)
    with torch.backends.cudnn.flags(enabled=False):
        ref, (hn, cn) = lstm(x, (h_in, c_in))
        gref = torch.autograd.grad((ref * wgt).sum(), [x] + list(lstm.parameters()))
    w_ih, w_hh = lstm.weight_ih_l0.detach(), lstm.weight_hh_l0.detach()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
    whht = w_hh.t().contiguous()
    new = lambda *s: torch.full(s, float("nan"), device=dev)
    gates, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    st = torch.cuda.current_stream().cuda_stream
    xd = x.detach().contiguous()
    if P == 0:
        gates.copy_(torch.addmm(bias, xd.view(T * B, H), w_ih.t()).view(T, B, 4 * H))
        rc = lib.kvae_lstm128_forward(T, B, 0, None, None, None, whht.data_ptr(), gates.data_ptr(),
                                      h.data_ptr(), c.data_ptr(), h_in[0].contiguous().data_ptr(),
                                      c_in[0].contiguous().data_ptr(), 1, st)
    else:
        rc = lib.kvae_lstm128_forward(T, B, P, xd.data_ptr(), w_ih.contiguous().data_ptr(),
                                      bias.data_ptr(), whht.data_ptr(), gates.data_ptr(),
                                      h.data_ptr(), c.data_ptr(), h_in[0].contiguous().data_ptr(),
                                      c_in[0].contiguous().data_ptr(), 1, st)
    assert rc == 0
    assert rel_err(h, ref) < 1e-5
    assert rel_err(h[-1], hn[0]) < 1e-5 and rel_err(c[-1], cn[0]) < 1e-5
    dG, dx = new(T, B, 4 * H), new(T, B, max(P, 1))
    part = new(lib.kvae_lstm128_blocks(B), 4 * H * (2 + P))
    rc = lib.kvae_lstm128_backward(T, B, P, xd.data_ptr() if P else None,
                                   w_ih.contiguous().data_ptr() if P else None, gates.data_ptr(),
                                   c.data_ptr(), c_in[0].contiguous().data_ptr(),
                                   wgt.contiguous().data_ptr(), w_hh.contiguous().data_ptr(),
                                   dG.data_ptr(), dx.data_ptr() if P else None, part.data_ptr(), st)
    assert rc == 0
    torch.cuda.synchronize()
    sums = part.sum(0)
    d2 = dG.view(T * B, 4 * H)
    h_prev = torch.cat([h_in, h[:-1]]).view(T * B, H)
    got = {"x": (d2 @ w_ih).view(T, B, -1) if P == 0 else dx,
           "weight_ih_l0": d2.t() @ xd.view(T * B, -1) if P == 0 else sums[8 * H:].view(4 * H, P),
           "weight_hh_l0": d2.t() @ h_prev, "bias_ih_l0": sums[:4 * H],
           "bias_hh_l0": sums[4 * H:8 * H]}
    names = ["x"] + [n for n, _ in lstm.named_parameters()]
    bad = [(n, rel_err(got[n], g)) for n, g in zip(names, gref) if not rel_err(got[n], g) < 1e-4]
    assert not bad, bad


@pytest.mark.parametrize("T,B,P,gates", [(1, 16 * 700 + 5, 2, 0), (1, 16 * 700, 0, 0),
                                         (3, 16 * 530 + 9, 0, 1), (2, 16 * 600, 3, 1)])
def test_resident_layer_many_tiles_per_workgroup_matches_torch(T, B, P, gates):
    """Short sequences over many rows launch one workgroup per compute unit, each working
    through several 16-row tiles with the recurrent matrix loaded once: outputs (and the
    stored gates) against torch.nn.LSTM from a carried (h, c), full and ragged last tile."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(T + B + P)
    dev = torch.device("cuda:0")
    H = 128
    in_dim = P if P > 0 else H
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev)
    x = torch.randn(T, B, in_dim, device=dev)
    h_in, c_in = 0.5 * torch.randn(1, B, H, device=dev), 0.5 * torch.randn(1, B, H, device=dev)
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        ref, (hn, cn) = lstm(x, (h_in, c_in))
    w_ih, w_hh = lstm.weight_ih_l0.detach(), lstm.weight_hh_l0.detach()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
    whht = w_hh.t().contiguous()
    new = lambda *s: torch.full(s, float("nan"), device=dev)
    g_buf, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    st = torch.cuda.current_stream().cuda_stream
    hi, ci = h_in[0].contiguous(), c_in[0].contiguous()
    if P == 0:
        g_buf.copy_(torch.addmm(bias, x.view(T * B, H), w_ih.t()).view(T, B, 4 * H))
        rc = lib.kvae_lstm128_forward(T, B, 0, None, None, None, whht.data_ptr(), g_buf.data_ptr(),
                                      h.data_ptr(), c.data_ptr(), hi.data_ptr(), ci.data_ptr(),
                                      gates, st)
    else:
        rc = lib.kvae_lstm128_forward(T, B, P, x.contiguous().data_ptr(),
                                      w_ih.contiguous().data_ptr(), bias.data_ptr(),
                                      whht.data_ptr(), g_buf.data_ptr() if gates else None,
                                      h.data_ptr(), c.data_ptr(), hi.data_ptr(), ci.data_ptr(),
                                      gates, st)
    assert rc == 0
    assert rel_err(h, ref) < 1e-5
    assert rel_err(h[-1], hn[0]) < 1e-5 and rel_err(c[-1], cn[0]) < 1e-5
    if gates:
        # the activated gates of the last step reproduce its cell update
        i_g, f_g, g_g, o_g = g_buf[-1].split(H, dim=-1)
        c_prev = c[-2] if T > 1 else ci
        assert rel_err(f_g * c_prev + i_g * g_g, c[-1]) < 1e-5
        assert rel_err(o_g * torch.tanh(c[-1]), h[-1]) < 1e-5


@pytest.mark.parametrize("M,K,N,with_bias", [(16 * 300, 128, 512, True), (16 * 300 + 7, 512, 128, False),
                                              (5, 128, 512, False), (81920, 512, 128, True),
                                              (1, 512, 128, False), (16 * 257 + 15, 128, 512, True)])
def test_rowstream_matmul_matches_float64(M, K, N, with_bias):
    """kvae_rowstream_matmul (the matrix resident in registers, rows streamed in 16-row tiles,
    split-bf16 products) against a float64 product: full and ragged last tiles, fewer tiles
    than workgroups, more tiles than workgroups."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(M + K)
    dev = torch.device("cuda:0")
    X = torch.randn(M, K, device=dev)
    W = torch.randn(K, N, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) if with_bias else None
    Y = torch.full((M, N), float("nan"), device=dev)
    rc = lib.kvae_rowstream_matmul(M, K, N, X.data_ptr(), W.data_ptr(),
                                   b.data_ptr() if with_bias else None, Y.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    ref = X.double() @ W.double() + (b.double() if with_bias else 0.0)
    assert rel_err(Y, ref) < 1e-5
    # every entry, not only the largest: the error of one entry against the size of its terms
    scale = X.double().abs() @ W.double().abs() + 1.0
    assert float(((Y.double() - ref).abs() / scale).max()) < 1e-5
    assert lib.kvae_rowstream_matmul(M, 64, 64, X.data_ptr(), W.data_ptr(), None, Y.data_ptr(),
                                     None) == -2   # KVAE_E_UNSUPPORTED


@pytest.mark.parametrize("T,B", [(20, 256), (3, 37), (2, 1), (7, 1000), (20, 4096)])
def test_wgrad_matches_float64(T, B):
    """kvae_lstm128_wgrad (dG^T H over T*B rows in slabs, split-bf16 products) against float64,
    with and without the one-step row offset; ragged slabs and fewer rows than one step."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(T * 1000 + B)
    dev = torch.device("cuda:0")
    H, R = 128, T * B
    E = 4 * H * H
    dG = torch.randn(R, 4 * H, device=dev)
    h0, h1 = torch.tanh(torch.randn(R, H, device=dev)), torch.tanh(torch.randn(R, H, device=dev))
    S = lib.kvae_lstm128_wgrad_slabs(R)
    assert 1 <= S <= 64
    slabs = torch.full((S, 3 * E), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for i, (hh, shift) in enumerate(((h0, B), (h0, 0), (h1, B))):
        assert lib.kvae_lstm128_wgrad(R, dG.data_ptr(), hh.data_ptr(), shift,
                                      slabs[:, i * E:].data_ptr(), 3 * E, st) == 0
    got = capi.column_sum(slabs).view(3, 4 * H, H)
    d64, a64, b64 = dG.double(), h0.double(), h1.double()
    ref = [d64[B:].t() @ a64[:R - B], d64.t() @ a64, d64[B:].t() @ b64[:R - B]]
    mag = [d64[B:].abs().t() @ a64[:R - B].abs(), d64.abs().t() @ a64.abs(),
           d64[B:].abs().t() @ b64[:R - B].abs()]
    for i in range(3):
        # each entry against the size of its terms (the sums themselves cancel to ~sqrt(R)):
        # a split product is good to ~1e-5 of its magnitude (5e-5 at worst), a sum of N of
        # them to ~1e-5/sqrt(N)
        err = float(((got[i].double() - ref[i]).abs() / (mag[i] + 1e-30)).max()) if R > B else \
            float(got[i].abs().max())
        assert err < (5e-5 if R < 4096 else 2e-6), (i, err)
    assert lib.kvae_lstm128_wgrad(R, dG.data_ptr(), h0.data_ptr(), 0, slabs.data_ptr(), E - 1,
                                  st) != 0


def test_column_sum_many_equals_separate_sums():
    """kvae_column_sum_many: several matrices in one launch give the bits of kvae_column_sum
    on each (ragged column counts, one matrix with a single row, float64 too)."""
    from kalman_vae_amd import _capi as capi
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    for dt in (torch.float32, torch.float64):
        xs = [torch.randn(256, 1024, dtype=dt, device=dev), torch.randn(37, 2051, dtype=dt, device=dev),
              torch.randn(1, 5, dtype=dt, device=dev), torch.randn(64, 70000, dtype=dt, device=dev)]
        got = capi.column_sum_many(xs)
        for x, g in zip(xs, got):
            assert torch.equal(g, capi.column_sum(x))
    # wide float matrices take a four-columns-per-lane form (16-byte loads); a copy at an
    # address that is not a multiple of 16 takes the one-column-per-lane form: same bits,
    # and both equal a float64 sum to rounding
    for rows, cols in ((64, 196608), (256, 1024), (37, 70000)):
        x = torch.randn(rows, cols, device=dev)
        shifted = torch.empty(rows * cols + 1, device=dev)[1:].view(rows, cols)
        shifted.copy_(x)
        assert shifted.data_ptr() % 16 != 0
        wide, narrow = capi.column_sum(x), capi.column_sum(shifted)
        assert torch.equal(wide, narrow)
        want = x.double().sum(0)
        assert float((wide.double() - want).abs().max()) < 1e-5 * float(x.double().abs().sum(0).max())
    assert len(capi.column_sum_many(xs[:1])) == 1
    with pytest.raises(RuntimeError):
        capi.column_sum_many(xs + xs[:1])


@pytest.mark.parametrize("T,B,P,K,lead", [(20, 70, 2, 3, True), (5, 33, 4, 2, False),
                                          (9, 256, 2, 4, True), (2, 1, 1, 3, False)])
def test_lstm_with_head_in_one_node_matches_separate_nodes(T, B, P, K, lead):
    """fused_lstm_head (the softmax head's adjoint inside the upper layer's backward kernel)
    against fused_lstm followed by softmax_head, and both against torch: weights, every
    parameter gradient, the state to continue from."""
    import kalman_vae_amd.lstm_function as lf
    torch.manual_seed(T * 7 + B + K)
    dev = torch.device("cuda:0")
    lstm = torch.nn.LSTM(P, 128, num_layers=2).to(dev)
    lin = torch.nn.Linear(128, K).to(dev)
    x = torch.randn(T, B, P, device=dev, requires_grad=True)
    w = torch.randn(T + (1 if lead else 0), B, K, device=dev)
    params = [x] + list(lstm.parameters()) + list(lin.parameters())
    fused = lf.fused_lstm_head(x, lstm, lin, lead)
    assert fused is not None
    a1, st1 = fused
    g1 = torch.autograd.grad((a1 * w).sum(), params)
    h, st2 = lf.fused_lstm(x, lstm)
    a2 = lf.softmax_head(h, lin, lead)
    g2 = torch.autograd.grad((a2 * w).sum(), params)
    with torch.backends.cudnn.flags(enabled=False):
        ref = torch.softmax(lin(lstm(x)[0]), -1)
    if lead:
        ref = torch.cat([torch.full((1, B, K), 1.0 / K, device=dev), ref])
    g3 = torch.autograd.grad((ref * w).sum(), params)
    assert torch.equal(a1, a2) and rel_err(a1, ref) < 1e-5
    for (hh1, cc1), (hh2, cc2) in [(st1.get(), st2.get())]:
        assert torch.equal(hh1, hh2) and torch.equal(cc1, cc2)
    names = ["x"] + [n for n, _ in lstm.named_parameters()] + ["linear.weight", "linear.bias"]
    bad = [(n, rel_err(a, b), rel_err(a, c)) for n, a, b, c in zip(names, g1, g2, g3)
           if not (rel_err(a, b) < 2e-5 and rel_err(a, c) < 1e-4)]
    assert not bad, bad


@pytest.mark.parametrize("P", [0, 1, 2])
@pytest.mark.parametrize("B,carried", [(16 * 300 + 5, True), (7, True), (16 * 2048, False),
                                       (1, False), (16 * 129, True)])
def test_layer_step_with_projection_in_kernel_matches_torch(B, carried, P):
    """kvae_lstm128_step_fused (one step of a layer over many rows, input projection in the
    kernel; the upper layer's hidden units cut over two workgroups) against one step of a
    torch LSTM layer in float64: fewer tiles than workgroups, many tiles per workgroup, ragged
    last tile, zero state; upper layer (P = 0, 128-wide input) and lower layer (narrow input)."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(B + P)
    dev = torch.device("cuda:0")
    H = 128
    in_dim = P if P > 0 else H
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev)
    x = torch.randn(1, B, in_dim, device=dev)
    h_in, c_in = 0.5 * torch.randn(1, B, H, device=dev), 0.5 * torch.randn(1, B, H, device=dev)
    ref64 = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev).double()
    ref64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        state = (h_in.double(), c_in.double()) if carried else None
        _, (hn, cn) = ref64(x.double(), state)
    w_ih = lstm.weight_ih_l0.detach()
    wx = w_ih.t().contiguous() if P == 0 else w_ih.contiguous()
    whht = lstm.weight_hh_l0.detach().t().contiguous()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
    h = torch.full((B, H), float("nan"), device=dev)
    c = torch.full((B, H), float("nan"), device=dev)
    hp, cp = h_in[0].contiguous(), c_in[0].contiguous()
    xs = x[0].contiguous()
    rc = lib.kvae_lstm128_step_fused(B, P, xs.data_ptr(), wx.data_ptr(),
                                     hp.data_ptr() if carried else None,
                                     cp.data_ptr() if carried else None, whht.data_ptr(),
                                     bias.data_ptr(), h.data_ptr(), c.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    # split products: ~1e-6 of the size of a gate sum's terms; the inputs here are N(0,1),
    # larger than the (-1,1) outputs of a lower layer, and the maximum is over up to 4 M entries
    assert rel_err(h, hn[0]) < 3e-5 and rel_err(c, cn[0]) < 3e-5
    assert lib.kvae_lstm128_step_fused(B, P, xs.data_ptr(), wx.data_ptr(), hp.data_ptr(), None,
                                       whht.data_ptr(), bias.data_ptr(), h.data_ptr(),
                                       c.data_ptr(), None) != 0
    assert lib.kvae_lstm128_step_fused(B, 3, xs.data_ptr(), wx.data_ptr(), None, None,
                                       whht.data_ptr(), bias.data_ptr(), h.data_ptr(),
                                       c.data_ptr(), None) == -2


@pytest.mark.parametrize("P", [0, 1, 2])
@pytest.mark.parametrize("B,carried", [(16 * 3 * 256, True), (16 * 3 * 256 + 5, True),
                                       (16 * 2 * 256, False), (16 * 200, True)])
def test_layer_step_every_entry_of_each_first_tile(B, carried, P):
    """kvae_lstm128_step_fused entry by entry against float64, with the first tile of every
    tile walker looked at apart: that tile takes the kernel's prologue path (accumulators and
    operands freshly placed), the one where a wrongly restored operand register shows.  A
    maximum over millions of entries hides a single wrong value of ordinary size, so every entry
    has its own bound; three launches on the same inputs must agree bit for bit."""
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    torch.manual_seed(7 * B + P)
    dev = torch.device("cuda:0")
    H = 128
    in_dim = P if P > 0 else H
    lstm = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev)
    x = torch.randn(B, in_dim, device=dev)
    hp, cp = 0.5 * torch.randn(B, H, device=dev), 0.5 * torch.randn(B, H, device=dev)
    ref64 = torch.nn.LSTM(in_dim, H, num_layers=1).to(dev).double()
    ref64.load_state_dict({k: v.double() for k, v in lstm.state_dict().items()})
    with torch.no_grad(), torch.backends.cudnn.flags(enabled=False):
        state = (hp[None].double(), cp[None].double()) if carried else None
        _, (hn, cn) = ref64(x[None].double(), state)
    hn, cn = hn[0], cn[0]
    w_ih = lstm.weight_ih_l0.detach()
    wx = w_ih.t().contiguous() if P == 0 else w_ih.contiguous()
    whht = lstm.weight_hh_l0.detach().t().contiguous()
    bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = (B + 15) // 16
    walkers = min(tiles, cus) if P > 0 else 8 * max(1, min(cus // 16, (tiles + 7) // 8))
    first_rows = min(B, 16 * walkers)              # tile k of walker w is tile w + k * walkers
    outs = []
    for _ in range(3):
        h = torch.full((B, H), float("nan"), device=dev)
        c = torch.full((B, H), float("nan"), device=dev)
        rc = lib.kvae_lstm128_step_fused(B, P, x.data_ptr(), wx.data_ptr(),
                                         hp.data_ptr() if carried else None,
                                         cp.data_ptr() if carried else None, whht.data_ptr(),
                                         bias.data_ptr(), h.data_ptr(), c.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append((h, c))
    for h, c in outs[1:]:
        assert torch.equal(h, outs[0][0]) and torch.equal(c, outs[0][1])
    h, c = outs[0]
    # split products: an absolute error of about 1e-6 times the terms of a gate sum (N(0,1)
    # inputs: up to ~1e-5 on a value in (-1, 1)); an operand dword taken from a stale register
    # moved entries by 1e-4 and more
    bound = 4e-5
    eh, ec = (h.double() - hn).abs(), (c.double() - cn).abs()
    wrong = ((eh > bound) | (ec > bound) | torch.isnan(eh) | torch.isnan(ec))
    where = wrong.nonzero()[:8].tolist()
    assert not where, (f"{int(wrong.sum())} wrong entries, {int(wrong[:first_rows].sum())} of them in "
                       f"first tiles; (row, unit): {where}")


@pytest.mark.parametrize("first", [1, 0])
def test_backward_multiplies_in_the_form_its_forward_used(first):
    """The product form is a process-wide setting; changing it between a forward pass and its
    backward pass must not mix the two forms: the gradient equals, bit for bit, the one of a
    run that never changed the setting, and the setting is left as the caller put it."""
    import kalman_vae_amd.lstm_function as lf
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    T, B, p, K = 6, 48, 2, 3
    lstm = torch.nn.LSTM(p, 128, num_layers=2).to(dev)
    lin = torch.nn.Linear(128, K).to(dev)
    x = torch.randn(T, B, p, device=dev, requires_grad=True)
    w = torch.randn(T + 1, B, K, device=dev)
    params = [x] + list(lstm.parameters()) + list(lin.parameters())
    try:
        assert lib.kvae_lstm128_set_precision(first) == 0
        a1, _ = lf.fused_lstm_head(x, lstm, lin, True)
        g_same = torch.autograd.grad((a1 * w).sum(), params)
        a2, _ = lf.fused_lstm_head(x, lstm, lin, True)
        assert lib.kvae_lstm128_set_precision(1 - first) == 0
        g_switched = torch.autograd.grad((a2 * w).sum(), params)
        assert lib.kvae_lstm128_get_precision() == 1 - first
        for a, b in zip(g_same, g_switched):
            assert torch.equal(a, b)
        h1, _ = lf.fused_lstm(x, lstm)
        assert lib.kvae_lstm128_set_precision(first) == 0
        g3 = torch.autograd.grad((h1 * h1).sum(), params[:-2])
        assert lib.kvae_lstm128_set_precision(1 - first) == 0
        h2, _ = lf.fused_lstm(x, lstm)
        g4 = torch.autograd.grad((h2 * h2).sum(), params[:-2])
        assert torch.equal(h1, h2)
        for a, b in zip(g3, g4):
            assert torch.equal(a, b)
    finally:
        lib.kvae_lstm128_set_precision(1)
