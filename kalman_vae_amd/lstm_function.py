"""Whole-sequence LSTM for the dynamics-parameter network with the recurrence
on HIP kernels (kalman_vae_amd/csrc/lstm_kernels.hip).

Same function as ``torch.nn.LSTM`` (gate order i,f,g,o, zero initial state) as
used by the reference's ``LSTMModel.forward`` (kvae/dynamics_parameter_network.py:
46-58).  Per layer the input projection of all steps and the weight gradients
are time-batched library GEMMs (torch.addmm / bmm -> rocBLAS); the T dependent
steps -- recurrent product fused with the cell update, and their adjoints -- are
the kernels.  MIOpen's LSTM spends ~12 ms per forward+backward at (T=20,
B=4096, H=128, 2 layers), most of it in one badly shaped weight-gradient GEMM;
this path needs no per-step library call at all.
"""
import torch

from . import _capi as capi


PERSISTENT = True  # use the one-launch kernels when the shape allows it
# two-layer H=128 network: one launch per LAYER with the recurrent matrix resident in
# registers (False: one launch for both layers, weights streamed from L2 every step)
RESIDENT = True


def supported(x, hidden_dim):
    return x.is_cuda and x.dtype == torch.float32 and hidden_dim in (32, 64, 128, 256)


def _splitk_tn(a, b):
    """a^T @ b for tall-skinny a (R,M), b (R,N) with R huge: a split over R into
    16-32 slabs keeps the rocBLAS call well shaped (one batched GEMM + a small sum;
    the plain transposed product is 3-4x slower at R = 81920)."""
    R = a.shape[0]
    # a slab count that is a multiple of 16 makes the batched GEMM's workgroup count a
    # multiple of the 256 CUs for the LSTM's (512 x 128) outputs (16 macro tiles per
    # slab): 78-80 us at R = 81920 against 84 us for 20 slabs and 96 us for 40
    for S in (32, 16, 64):
        if R % S == 0 and R // S >= 512:
            chunk = R // S
            break
    else:
        chunk = 512
        while chunk * 32 <= R:
            chunk *= 2
        S = R // chunk
    if S < 4:
        return a.t() @ b
    slabs = torch.bmm(a[:S * chunk].view(S, chunk, -1).transpose(1, 2),
                      b[:S * chunk].view(S, chunk, -1))
    main = capi.column_sum(slabs.view(S, -1)).view(slabs.shape[1:])
    if S * chunk < R:
        main = main + a[S * chunk:].t() @ b[S * chunk:]
    return main


def _slab_plan(R):
    """(slab count, rows per slab) _splitk_tn uses for R rows, or None when it would not
    split."""
    for S in (32, 16, 64):
        if R % S == 0 and R // S >= 512:
            return S, R // S
    return None


def _splitk_tn_many(pairs):
    """[a_i^T @ b_i] for several tall-skinny pairs whose products have the same shape:
    when every pair splits into the same number of slabs without a remainder, the
    batched GEMMs write into one buffer and ONE launch sums the slabs of all of them
    (each launch of a dependent chain costs ~5 us whatever its size)."""
    plans = [_slab_plan(a.shape[0]) for a, _ in pairs]
    shapes = {(a.shape[1], b.shape[1]) for a, b in pairs}
    if None in plans or len({pl[0] for pl in plans}) != 1 or len(shapes) != 1 or len(pairs) < 2:
        return [_splitk_tn(a, b) for a, b in pairs]
    S = plans[0][0]
    (M, N), n = next(iter(shapes)), len(pairs)
    slabs = torch.empty(S, n, M, N, dtype=pairs[0][0].dtype, device=pairs[0][0].device)
    for i, ((a, b), (_, chunk)) in enumerate(zip(pairs, plans)):
        torch.bmm(a.view(S, chunk, M).transpose(1, 2), b.view(S, chunk, N), out=slabs[:, i])
    sums = capi.column_sum(slabs.view(S, -1)).view(n, M, N)
    return [sums[i] for i in range(n)]


class FusedLSTMFunction(torch.autograd.Function):
    """(x (T,B,in), w_ih_0, w_hh_0, b_ih_0, b_hh_0, w_ih_1, ...) ->
    (top-layer h (T,B,H), h_n (L,B,H), c_n (L,B,H))."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = capi.load()
        T, B, _ = x.shape
        L = len(params) // 4
        H = params[1].shape[1]
        stream = capi.current_stream_ptr()
        saved, inp = [], x.contiguous()
        h_n, c_n = [], []
        for l in range(L):
            w_ih, w_hh, b_ih, b_hh = params[4 * l:4 * l + 4]
            gates = torch.addmm(b_ih + b_hh, inp.reshape(T * B, -1), w_ih.t()).view(T, B, 4 * H)
            h = torch.empty(T, B, H, dtype=x.dtype, device=x.device)
            c = torch.empty_like(h)
            capi.check(lib.kvae_lstm_layer_forward(T, B, H, gates.data_ptr(),
                                                   w_hh.t().contiguous().data_ptr(),
                                                   h.data_ptr(), c.data_ptr(), stream))
            saved += [inp, gates, h, c]
            h_n.append(h[-1])
            c_n.append(c[-1])
            inp = h
        ctx.save_for_backward(*saved, *params)
        ctx.L, ctx.H = L, H
        h_n, c_n = torch.stack(h_n), torch.stack(c_n)
        ctx.mark_non_differentiable(h_n, c_n)
        ctx.set_materialize_grads(False)
        return inp, h_n, c_n

    @staticmethod
    def backward(ctx, dh_top, _dhn, _dcn):  # the state outputs carry no gradient
        lib = capi.load()
        L, H = ctx.L, ctx.H
        tensors = ctx.saved_tensors
        saved, params = tensors[:4 * L], tensors[4 * L:]
        T, B, _ = saved[2].shape
        stream = capi.current_stream_ptr()
        dh = dh_top.contiguous().clone()
        dc_work = torch.empty(B, H, dtype=dh.dtype, device=dh.device)
        grads = [None] * (4 * L)
        dx = None
        for l in reversed(range(L)):
            inp, gates, h, c = saved[4 * l:4 * l + 4]
            w_ih, w_hh = params[4 * l], params[4 * l + 1]
            dG = torch.empty_like(gates)
            capi.check(lib.kvae_lstm_layer_backward(T, B, H, gates.data_ptr(), c.data_ptr(),
                                                    dh.data_ptr(), w_hh.contiguous().data_ptr(),
                                                    dG.data_ptr(), dc_work.data_ptr(), stream))
            dG2 = dG.view(T * B, 4 * H)
            grads[4 * l] = _splitk_tn(dG2, inp.reshape(T * B, -1))
            if T > 1:
                grads[4 * l + 1] = _splitk_tn(dG2[B:], h[:-1].reshape((T - 1) * B, H))
            else:
                grads[4 * l + 1] = torch.zeros_like(w_hh)
            db = dG2.sum(0)
            grads[4 * l + 2] = db
            grads[4 * l + 3] = db
            below = (dG2 @ w_ih).view(T, B, -1)
            if l > 0:
                dh = below
            else:
                dx = below
        return (dx, *grads)


def _lstm2_operands(w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1):
    """(Whh0T, Wih1T, Whh1T, b0, b1) for the persistent forward kernel, built by one
    launch into one buffer (views of it are returned)."""
    lib = capi.load()
    H, G = 128, 512
    buf = torch.empty(lib.kvae_lstm2_prepared_elems(), dtype=w_hh0.dtype, device=w_hh0.device)
    capi.check(lib.kvae_lstm2_prepare(
        w_hh0.contiguous().data_ptr(), w_ih1.contiguous().data_ptr(),
        w_hh1.contiguous().data_ptr(), b_ih0.contiguous().data_ptr(),
        b_hh0.contiguous().data_ptr(), b_ih1.contiguous().data_ptr(),
        b_hh1.contiguous().data_ptr(), buf.data_ptr(), capi.current_stream_ptr()))
    mats = buf[:3 * H * G].view(3, H, G)
    return mats[0], mats[1], mats[2], buf[3 * H * G:3 * H * G + G], buf[3 * H * G + G:]


class FusedLSTM2Function(torch.autograd.Function):
    """Two layers, hidden size 128, input dimension <= 4: the persistent kernels
    (one launch for all T steps of both layers, forward and backward)."""

    @staticmethod
    def forward(ctx, x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1):
        lib = capi.load()
        T, B, P = x.shape
        H = 128
        x = x.contiguous()
        new = lambda *s: torch.empty(s, dtype=x.dtype, device=x.device)
        gates0, h0, c0 = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
        gates1, h1, c1 = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
        whh0t, wih1t, whh1t, b0, b1 = _lstm2_operands(w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1,
                                                      b_hh1)
        keep = (w_ih0.contiguous(), b0, whh0t, wih1t, whh1t, b1)
        capi.check(lib.kvae_lstm2_forward(
            T, B, P, x.data_ptr(), *(k.data_ptr() for k in keep), gates0.data_ptr(),
            h0.data_ptr(), c0.data_ptr(), gates1.data_ptr(), h1.data_ptr(), c1.data_ptr(),
            None, None, capi.current_stream_ptr()))
        ctx.save_for_backward(x, gates0, h0, c0, gates1, h1, c1, w_ih0, w_hh0, w_ih1, w_hh1)
        # the lower layer's h and both cell sequences go out too (not differentiable):
        # their last steps are the state a later call continues from, and the caller
        # assembles it only if such a call comes
        ctx.mark_non_differentiable(h0, c0, c1)
        ctx.set_materialize_grads(False)
        return h1, h0, c0, c1

    @staticmethod
    def backward(ctx, g_h1, _dh0, _dc0, _dc1):
        lib = capi.load()
        x, gates0, h0, c0, gates1, h1, c1, w_ih0, w_hh0, w_ih1, w_hh1 = ctx.saved_tensors
        T, B, P = x.shape
        H = 128
        g_h1 = g_h1.contiguous()
        dG0, dG1 = torch.empty_like(gates0), torch.empty_like(gates1)
        dx = torch.empty_like(x)
        partial = torch.empty(lib.kvae_lstm2_backward_blocks(B), 4 * H * (4 + P),
                              dtype=x.dtype, device=x.device)
        wih0, whh0, wih1, whh1 = (w.contiguous() for w in (w_ih0, w_hh0, w_ih1, w_hh1))
        capi.check(lib.kvae_lstm2_backward(
            T, B, P, g_h1.data_ptr(), x.data_ptr(), gates0.data_ptr(), c0.data_ptr(),
            gates1.data_ptr(), c1.data_ptr(), wih0.data_ptr(), whh0.data_ptr(), wih1.data_ptr(),
            whh1.data_ptr(), dG0.data_ptr(), dG1.data_ptr(), dx.data_ptr(), partial.data_ptr(),
            capi.current_stream_ptr()))
        sums = capi.column_sum(partial)
        # each bias sum comes out twice: separate memory for the gradients of bias_ih and
        # bias_hh (the same tensor returned twice would be copied by autograd)
        db0, db0b, db1, db1b = (sums[i * 4 * H:(i + 1) * 4 * H] for i in range(4))
        dw_ih0 = sums[16 * H:].view(4 * H, P)
        d0, d1 = dG0.view(T * B, 4 * H), dG1.view(T * B, 4 * H)
        zero = lambda w: torch.zeros_like(w)
        if T > 1:
            dw_hh0, dw_ih1, dw_hh1 = _splitk_tn_many([
                (d0[B:], h0[:-1].reshape(-1, H)), (d1, h0.reshape(T * B, H)),
                (d1[B:], h1[:-1].reshape(-1, H))])
        else:
            dw_hh0, dw_ih1, dw_hh1 = zero(w_hh0), _splitk_tn(d1, h0.reshape(T * B, H)), zero(w_hh1)
        return dx, dw_ih0, dw_hh0, db0, db0b, dw_ih1, dw_hh1, db1, db1b


def _rows_matmul(x2, w, bias, out):
    """out = x2 @ w (+ bias) for the two tall products around the resident layer kernels:
    the library's row-streaming kernel (w resident in registers, x2 and out through HBM
    once) when the product forms agree with the layer kernels', a library GEMM otherwise."""
    lib = capi.load()
    if lib.kvae_lstm128_get_precision() == 1 and x2.dtype == torch.float32:
        w = w.contiguous()
        capi.check(lib.kvae_rowstream_matmul(
            x2.shape[0], w.shape[0], w.shape[1], x2.data_ptr(), w.data_ptr(),
            bias.data_ptr() if bias is not None else None, out.data_ptr(),
            capi.current_stream_ptr()))
        return out
    if bias is not None:
        return torch.addmm(bias, x2, w, out=out)
    return torch.mm(x2, w, out=out)


def _resident_forward(x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1):
    """Both layers through kvae_lstm128_forward: (gates0, h0, c0, gates1, h1, c1)."""
    lib = capi.load()
    T, B, P = x.shape
    H = 128
    new = lambda *s: torch.empty(s, dtype=x.dtype, device=x.device)
    gates0, h0, c0 = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    gates1, h1, c1 = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    whh0t, wih1t, whh1t, b0, b1 = _lstm2_operands(w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1,
                                                  b_hh1)
    wih0 = w_ih0.contiguous()
    stream = capi.current_stream_ptr()
    if P <= 4:
        capi.check(lib.kvae_lstm128_forward(
            T, B, P, x.data_ptr(), wih0.data_ptr(), b0.data_ptr(), whh0t.data_ptr(),
            gates0.data_ptr(), h0.data_ptr(), c0.data_ptr(), None, None, 1, stream))
    else:
        # a wide input: its projection of all steps is one library GEMM, like layer 1's
        torch.addmm(b0, x.view(T * B, P), wih0.t(), out=gates0.view(T * B, 4 * H))
        capi.check(lib.kvae_lstm128_forward(
            T, B, 0, None, None, None, whh0t.data_ptr(), gates0.data_ptr(), h0.data_ptr(),
            c0.data_ptr(), None, None, 1, stream))
    # layer 1's input projection of all steps: (T*B,128) x (128,512) + bias
    _rows_matmul(h0.view(T * B, H), wih1t, b1, gates1.view(T * B, 4 * H))
    capi.check(lib.kvae_lstm128_forward(
        T, B, 0, None, None, None, whh1t.data_ptr(), gates1.data_ptr(), h1.data_ptr(),
        c1.data_ptr(), None, None, 1, stream))
    return gates0, h0, c0, gates1, h1, c1


def _resident_backward(x, gates0, h0, c0, gates1, h1, c1, w_ih0, w_hh0, w_ih1, w_hh1, g_h1=None,
                       head=None):
    """Adjoint of _resident_forward.  Either g_h1 (T,B,H), the gradient w.r.t. the upper
    layer's h, or head = (alpha (T,B,K), g_alpha (T,B,K), W (K,H)): the softmax head's adjoint
    runs inside the upper layer's kernel, and (dW_head, db_head) are appended to the result."""
    lib = capi.load()
    T, B, P = x.shape
    H = 128
    dG0, dG1 = torch.empty_like(gates0), torch.empty_like(gates1)
    dx = torch.empty_like(x)
    blocks = lib.kvae_lstm128_blocks(B)
    part1 = torch.empty(blocks, 4 * H * 2, dtype=x.dtype, device=x.device)
    Pk = P if P <= 4 else 0          # input dimension the layer-0 kernel handles itself
    part0 = torch.empty(blocks, 4 * H * (2 + Pk), dtype=x.dtype, device=x.device)
    wih0, whh0, wih1, whh1 = (w.contiguous() for w in (w_ih0, w_hh0, w_ih1, w_hh1))
    stream = capi.current_stream_ptr()
    parth = None
    if head is None:
        capi.check(lib.kvae_lstm128_backward(
            T, B, 0, None, None, gates1.data_ptr(), c1.data_ptr(), None, g_h1.data_ptr(),
            whh1.data_ptr(), dG1.data_ptr(), None, part1.data_ptr(), stream))
    else:
        alpha, g_alpha, w_head = head
        K = w_head.shape[0]
        parth = torch.empty(blocks, K * (H + 1), dtype=x.dtype, device=x.device)
        capi.check(lib.kvae_lstm128_backward_head(
            T, B, K, gates1.data_ptr(), c1.data_ptr(), None, alpha.data_ptr(), g_alpha.data_ptr(),
            w_head.data_ptr(), whh1.data_ptr(), dG1.data_ptr(), part1.data_ptr(),
            parth.data_ptr(), stream))
    d0, d1 = dG0.view(T * B, 4 * H), dG1.view(T * B, 4 * H)
    # gradient w.r.t. layer 0's h through layer 1's input product
    g_h0 = _rows_matmul(d1, wih1, None, torch.empty(T * B, H, dtype=x.dtype,
                                                    device=x.device)).view(T, B, H)
    if Pk:
        capi.check(lib.kvae_lstm128_backward(
            T, B, P, x.data_ptr(), wih0.data_ptr(), gates0.data_ptr(), c0.data_ptr(), None,
            g_h0.data_ptr(), whh0.data_ptr(), dG0.data_ptr(), dx.data_ptr(),
            part0.data_ptr(), stream))
    else:
        capi.check(lib.kvae_lstm128_backward(
            T, B, 0, None, None, gates0.data_ptr(), c0.data_ptr(), None, g_h0.data_ptr(),
            whh0.data_ptr(), dG0.data_ptr(), None, part0.data_ptr(), stream))
    split = T > 1 and x.dtype == torch.float32 and lib.kvae_lstm128_get_precision() == 1
    sums = [part1, part0] + ([parth] if parth is not None else [])
    if split:
        # dG0^T h0_{t-1}, dG1^T h0_t, dG1^T h1_{t-1}: one launch each into one buffer of
        # per-slab results; ONE launch then sums the slabs of all three and the per-workgroup
        # bias / input-weight (/ head) partial sums
        R, E = T * B, 4 * H * H
        S = lib.kvae_lstm128_wgrad_slabs(R)
        slabs = torch.empty(S, 3 * E, dtype=x.dtype, device=x.device)
        for i, (dg, hh, shift) in enumerate(((d0, h0, B), (d1, h0, 0), (d1, h1, B))):
            capi.check(lib.kvae_lstm128_wgrad(R, dg.data_ptr(), hh.data_ptr(), shift,
                                              slabs[:, i * E:].data_ptr(), 3 * E, stream))
        sums.append(slabs)
    sums = capi.column_sum_many(sums)
    s1, s0 = sums[0], sums[1]
    db1, db1b = s1[:4 * H], s1[4 * H:]
    db0, db0b = s0[:4 * H], s0[4 * H:8 * H]
    if Pk:
        dw_ih0 = s0[8 * H:].view(4 * H, P)
    else:
        torch.mm(d0, wih0, out=dx.view(T * B, P))
        dw_ih0 = _splitk_tn(d0, x.view(T * B, P))
    zero = lambda w: torch.zeros_like(w)
    if split:
        dw_hh0, dw_ih1, dw_hh1 = sums[-1].view(3, 4 * H, H).unbind(0)
    elif T > 1:
        dw_hh0, dw_ih1, dw_hh1 = _splitk_tn_many([
            (d0[B:], h0[:-1].reshape(-1, H)), (d1, h0.reshape(T * B, H)),
            (d1[B:], h1[:-1].reshape(-1, H))])
    else:
        dw_hh0, dw_ih1, dw_hh1 = zero(w_hh0), _splitk_tn(d1, h0.reshape(T * B, H)), zero(w_hh1)
    out = (dx, dw_ih0, dw_hh0, db0, db0b, dw_ih1, dw_hh1, db1, db1b)
    if parth is not None:
        sh = sums[2]
        K = head[2].shape[0]
        out += (sh[:K * H].view(K, H), sh[K * H:])
    return out


class _ProductForm:
    """The library's product form (0: exact fp32 matrix instructions, 1: three bf16
    instructions per product) is a process-wide setting read at every launch.  A backward pass
    must multiply in the form its forward pass used: the saved gates came from that form.  The
    autograd nodes below record the form at forward time and run their backward launches
    inside ``with _ProductForm(form)``, which sets it for the launches and puts back whatever
    was set before (launches read the setting on the host, when they are issued)."""

    def __init__(self, form):
        self.form, self.prev = int(form), None

    @staticmethod
    def current():
        return int(capi.load().kvae_lstm128_get_precision())

    def __enter__(self):
        lib = capi.load()
        self.prev = int(lib.kvae_lstm128_get_precision())
        if self.prev != self.form:
            lib.kvae_lstm128_set_precision(self.form)
        return self

    def __exit__(self, *exc):
        if self.prev != self.form:
            capi.load().kvae_lstm128_set_precision(self.prev)
        return False


class ResidentLSTM2Function(torch.autograd.Function):
    """Two layers, hidden size 128, one launch per layer and direction with the layer's
    recurrent matrix resident in registers (kvae_lstm128_forward / _backward).  Layer 1's
    input product over all steps, the product that carries the gradient from layer 1 to
    layer 0 and the three recurrent / input weight gradients are the library's own
    time-batched kernels (library GEMMs with the exact-fp32 product form)."""

    @staticmethod
    def forward(ctx, x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1):
        x = x.contiguous()
        ctx.product_form = _ProductForm.current()
        gates0, h0, c0, gates1, h1, c1 = _resident_forward(x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1,
                                                           w_hh1, b_ih1, b_hh1)
        ctx.save_for_backward(x, gates0, h0, c0, gates1, h1, c1, w_ih0, w_hh0, w_ih1, w_hh1)
        ctx.mark_non_differentiable(h0, c0, c1)
        ctx.set_materialize_grads(False)
        return h1, h0, c0, c1

    @staticmethod
    def backward(ctx, g_h1, _dh0, _dc0, _dc1):
        with _ProductForm(ctx.product_form):
            return _resident_backward(*ctx.saved_tensors, g_h1=g_h1.contiguous())


class ResidentLSTM2HeadFunction(torch.autograd.Function):
    """ResidentLSTM2Function followed by the softmax head alpha = softmax(h1 W^T + b) as ONE
    autograd node: the head's adjoint runs inside the upper layer's backward kernel
    (kvae_lstm128_backward_head), which neither reads a gradient w.r.t. h1 nor needs the
    head's own backward launch and reduction.  K in 2..4, either product form (the backward
    pass runs in the one the forward pass used)."""

    @staticmethod
    def forward(ctx, x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, w_head, b_head,
                lead):
        lib = capi.load()
        x = x.contiguous()
        T, B, _ = x.shape
        H, K = 128, w_head.shape[0]
        ctx.product_form = _ProductForm.current()
        gates0, h0, c0, gates1, h1, c1 = _resident_forward(x, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1,
                                                           w_hh1, b_ih1, b_hh1)
        wc = w_head.contiguous()
        lead_rows = B if lead else 0
        out = torch.empty(lead_rows + T * B, K, dtype=x.dtype, device=x.device)
        capi.check(lib.kvae_softmax_head_forward(T * B, H, K, h1.data_ptr(), wc.data_ptr(),
                                                 b_head.contiguous().data_ptr(), out.data_ptr(),
                                                 lead_rows, capi.current_stream_ptr()))
        ctx.save_for_backward(x, gates0, h0, c0, gates1, h1, c1, w_ih0, w_hh0, w_ih1, w_hh1, wc,
                              out)
        ctx.lead_rows = lead_rows
        ctx.mark_non_differentiable(h1, h0, c0, c1)
        ctx.set_materialize_grads(False)
        return out.view(T + (1 if lead else 0), B, K), h1, h0, c0, c1

    @staticmethod
    def backward(ctx, g_alpha, _dh1, _dh0, _dc0, _dc1):
        if g_alpha is None:
            return (None,) * 12
        *saved, wc, out = ctx.saved_tensors
        x = saved[0]
        T, B, _ = x.shape
        K = wc.shape[0]
        g = g_alpha.contiguous().view(-1, K)[ctx.lead_rows:]
        with _ProductForm(ctx.product_form):
            grads = _resident_backward(*saved, head=(out[ctx.lead_rows:], g, wc))
        return grads + (None,)


class SoftmaxHeadFunction(torch.autograd.Function):
    """softmax(h @ W^T + b) over the last axis: h (..., H), W (K, H), b (K).  With
    lead = True and h (T,B,H) the result is (T+1,B,K) whose first step is the
    uniform 1/K (written by the same launch)."""

    @staticmethod
    def forward(ctx, h, W, b, lead):
        lib = capi.load()
        H, K = h.shape[-1], W.shape[0]
        h2, Wc = h.contiguous().view(-1, H), W.contiguous()
        lead_rows = h.shape[1] if lead else 0
        out = torch.empty(lead_rows + h2.shape[0], K, dtype=h.dtype, device=h.device)
        capi.check(lib.kvae_softmax_head_forward(h2.shape[0], H, K, h2.data_ptr(), Wc.data_ptr(),
                                                 b.contiguous().data_ptr(), out.data_ptr(),
                                                 lead_rows, capi.current_stream_ptr()))
        alpha = out[lead_rows:]
        ctx.save_for_backward(h2, Wc, alpha)
        ctx.lead, ctx.lead_rows = h.shape[:-1], lead_rows
        if lead:
            return out.view(h.shape[0] + 1, h.shape[1], K)
        return out.view(*h.shape[:-1], K)

    @staticmethod
    def backward(ctx, g):
        lib = capi.load()
        h2, Wc, alpha = ctx.saved_tensors
        R, H = h2.shape
        K = Wc.shape[0]
        g2 = g.contiguous().view(-1, K)[ctx.lead_rows:]
        d_h = torch.empty_like(h2)
        partial = torch.empty(lib.kvae_softmax_head_backward_blocks(R), K * (H + 1),
                              dtype=h2.dtype, device=h2.device)
        capi.check(lib.kvae_softmax_head_backward(R, H, K, h2.data_ptr(), Wc.data_ptr(),
                                                  alpha.data_ptr(), g2.data_ptr(), d_h.data_ptr(),
                                                  partial.data_ptr(), capi.current_stream_ptr()))
        sums = capi.column_sum(partial)
        return d_h.view(*ctx.lead, H), sums[:K * H].view(K, H), sums[K * H:], None


class LinearFunction(torch.autograd.Function):
    """x (..., I) @ W^T + b with the bias gradient summed by the package's
    deterministic column sum.  torch.nn.Linear's own bias gradient is a many-row
    reduction whose kernel zeroes a semaphore with a memset node, which keeps a step
    that contains it from being replayed as a HIP graph on this ROCm version."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        return torch.nn.functional.linear(x, W, b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        d_x = (g2 @ W).view(x.shape) if ctx.needs_input_grad[0] else None
        d_W = None
        if ctx.needs_input_grad[1]:
            if x2.shape[1] <= 8:
                # a few input features (the network's first layer sees a_dim of them): a
                # GEMM with so narrow an operand is badly shaped, one scaled column sum
                # per feature is not
                g2c = g2.contiguous()
                d_W = torch.stack([capi.column_sum_tall(g2c * x2[:, q:q + 1])
                                   for q in range(x2.shape[1])], dim=1)
            else:
                d_W = _splitk_tn(g2, x2)
        d_b = capi.column_sum_tall(g2.contiguous()) if ctx.needs_input_grad[2] else None
        return d_x, d_W, d_b


def linear(x, layer):
    """layer(x) for a torch.nn.Linear; on a GPU in float32/float64 with a bias the
    gradient path avoids library reductions (see LinearFunction)."""
    if x.is_cuda and layer.bias is not None and x.dtype in (torch.float32, torch.float64):
        return LinearFunction.apply(x, layer.weight, layer.bias)
    return layer(x)


def head_supported(h, linear):
    return (h.is_cuda and h.dtype == torch.float32 and linear.in_features in (32, 64, 128, 256)
            and linear.out_features <= 8 and linear.bias is not None)


def softmax_head(h, linear, lead=False):
    """softmax(linear(h), dim=-1) for a torch.nn.Linear with few outputs: the HIP
    head kernels on a GPU in float32, plain torch otherwise.  lead=True (h (T,B,H))
    prepends one step of uniform weights 1/K: result (T+1,B,K)."""
    if head_supported(h, linear):
        return SoftmaxHeadFunction.apply(h, linear.weight, linear.bias, lead)
    alpha = torch.softmax(linear(h), dim=-1)
    if lead:
        first = torch.full((1,) + tuple(alpha.shape[1:]), 1.0 / alpha.shape[-1], dtype=alpha.dtype,
                           device=alpha.device)
        alpha = torch.cat([first, alpha], dim=0)
    return alpha


def persistent_supported(x, lstm):
    return (supported(x, lstm.hidden_size) and lstm.num_layers == 2 and lstm.hidden_size == 128
            and x.shape[2] <= 4 and not lstm.bidirectional)


@torch.no_grad()
def lstm2_nograd(x, lstm, hidden=None):
    """Inference through the persistent two-layer kernel from a carried state:
    x (T,B,in), hidden = (h (2,B,H), c (2,B,H)) or None.  Returns (top-layer h
    (T,B,H), (h_n, c_n)).  This is what the step-by-step (masked) filter calls with
    T = 1 at every step; no gates are stored and no autograd graph is recorded."""
    lib = capi.load()
    T, B, P = x.shape
    H = 128
    x = x.contiguous()
    new = lambda *s: torch.empty(s, dtype=x.dtype, device=x.device)
    if T == 1:
        # one step: the kernel writes the two layers' h and c straight into the
        # (2,B,H) state tensors the next call starts from (no stacking copies)
        h_n, c_n = new(2, B, H), new(2, B, H)
        h0, h1, c0, c1 = h_n[0:1], h_n[1:2], c_n[0:1], c_n[1:2]
    else:
        h_n = c_n = None
        h0, c0, h1, c1 = new(T, B, H), new(T, B, H), new(T, B, H), new(T, B, H)
    w = lambda n: getattr(lstm, n)
    whh0t, wih1t, whh1t, b0, b1 = _lstm2_operands(
        w("weight_hh_l0"), w("bias_ih_l0"), w("bias_hh_l0"), w("weight_ih_l1"), w("weight_hh_l1"),
        w("bias_ih_l1"), w("bias_hh_l1"))
    keep = (w("weight_ih_l0").contiguous(), b0, whh0t, wih1t, whh1t, b1)
    h_init = c_init = None
    if hidden is not None:
        h_init, c_init = hidden[0].contiguous(), hidden[1].contiguous()
    if RESIDENT:
        # one launch per layer with the layer's recurrent matrix resident in registers (a
        # one-step call over many rows works through them with one workgroup per compute
        # unit), layer 1's input projection as one library GEMM in between
        stream = capi.current_stream_ptr()
        init = lambda t, l: None if t is None else t[l].data_ptr()
        if T == 1 and x.dtype == torch.float32 and lib.kvae_lstm128_get_precision() == 1:
            # one step over many rows: each layer as one launch with its input projection in
            # the kernel (the upper layer's (B,512) projection never goes through memory) and
            # several tiles of rows in flight per compute unit
            if P <= 2:
                capi.check(lib.kvae_lstm128_step_fused(
                    B, P, x.data_ptr(), keep[0].data_ptr(), init(h_init, 0), init(c_init, 0),
                    whh0t.data_ptr(), b0.data_ptr(), h0.data_ptr(), c0.data_ptr(), stream))
            else:
                capi.check(lib.kvae_lstm128_forward(
                    T, B, P, x.data_ptr(), keep[0].data_ptr(), b0.data_ptr(), whh0t.data_ptr(),
                    None, h0.data_ptr(), c0.data_ptr(), init(h_init, 0), init(c_init, 0), 0,
                    stream))
            capi.check(lib.kvae_lstm128_step_fused(
                B, 0, h0.data_ptr(), wih1t.data_ptr(), init(h_init, 1), init(c_init, 1),
                whh1t.data_ptr(), b1.data_ptr(), h1.data_ptr(), c1.data_ptr(), stream))
        else:
            capi.check(lib.kvae_lstm128_forward(
                T, B, P, x.data_ptr(), keep[0].data_ptr(), b0.data_ptr(), whh0t.data_ptr(), None,
                h0.data_ptr(), c0.data_ptr(), init(h_init, 0), init(c_init, 0), 0, stream))
            gates1 = _rows_matmul(h0.view(T * B, H), wih1t, b1, new(T * B, 4 * H))
            capi.check(lib.kvae_lstm128_forward(
                T, B, 0, None, None, None, whh1t.data_ptr(), gates1.data_ptr(), h1.data_ptr(),
                c1.data_ptr(), init(h_init, 1), init(c_init, 1), 0, stream))
        if T == 1:
            return h1, (h_n, c_n)
        return h1, (torch.stack([h0[-1], h1[-1]]), torch.stack([c0[-1], c1[-1]]))
    capi.check(lib.kvae_lstm2_forward(
        T, B, P, x.data_ptr(), *(k.data_ptr() for k in keep), None, h0.data_ptr(), c0.data_ptr(),
        None, h1.data_ptr(), c1.data_ptr(), capi.ptr(h_init), capi.ptr(c_init),
        capi.current_stream_ptr()))
    if T == 1:
        return h1, (h_n, c_n)
    return h1, (torch.stack([h0[-1], h1[-1]]), torch.stack([c0[-1], c1[-1]]))


class LSTM2HeadStepper:
    """softmax(linear(lstm(x_t))) one time step at a time from a carried state, for the
    step-by-step masked filter when no gradient is recorded (kvae/state_space_model.py:
    279-282 calls the network once per step): the weight operands are prepared once, the
    carried (h, c) alternate between two buffers, and a step is three launches (lower
    layer, upper layer, head) that write the weights into the caller's tensor.
    ``LSTM2HeadStepper.make`` returns None when this form does not apply."""

    @staticmethod
    def make(x, lstm, linear, hidden=None):
        """x: one step's input (B, P), float32 on a GPU."""
        if not (RESIDENT and x.dim() == 2 and not torch.is_grad_enabled()
                and persistent_supported(x.unsqueeze(0), lstm) and x.shape[1] <= 2
                and x.dtype == torch.float32 and head_supported(x.new_empty(1, 128), linear)
                and capi.load().kvae_lstm128_get_precision() == 1):
            return None
        return LSTM2HeadStepper(x, lstm, linear, hidden)

    def __init__(self, x, lstm, linear, hidden):
        H = 128
        self.B, self.P, self.K = x.shape[0], x.shape[1], linear.out_features
        w = lambda n: getattr(lstm, n).detach()
        self.whh0t, self.wih1t, self.whh1t, self.b0, self.b1 = _lstm2_operands(
            w("weight_hh_l0"), w("bias_ih_l0"), w("bias_hh_l0"), w("weight_ih_l1"),
            w("weight_hh_l1"), w("bias_ih_l1"), w("bias_hh_l1"))
        self.wih0 = w("weight_ih_l0").contiguous()
        self.Wc, self.bc = linear.weight.detach().contiguous(), linear.bias.detach().contiguous()
        new = lambda: torch.empty(2, self.B, H, dtype=x.dtype, device=x.device)
        self.state = [(new(), new()), (new(), new())]
        self.now = None          # index of the buffer pair that holds the carried state
        if hidden is not None:
            self.state[0][0].copy_(hidden[0])
            self.state[0][1].copy_(hidden[1])
            self.now = 0

    @torch.no_grad()
    def step(self, x, out):
        """x (B,P) -> out (B,K), both contiguous; advances the carried state."""
        if not (x.is_contiguous() and out.is_contiguous() and x.shape == (self.B, self.P)
                and out.shape == (self.B, self.K) and x.dtype == out.dtype == torch.float32):
            raise ValueError("LSTM2HeadStepper.step: x (B,P) and out (B,K) must be contiguous float32")
        lib, stream = capi.load(), capi.current_stream_ptr()
        nxt = 0 if self.now is None else self.now ^ 1
        h, c = self.state[nxt]
        prev = lambda l, i: None if self.now is None else self.state[self.now][i][l].data_ptr()
        capi.check(lib.kvae_lstm128_step_fused(
            self.B, self.P, x.data_ptr(), self.wih0.data_ptr(), prev(0, 0), prev(0, 1),
            self.whh0t.data_ptr(), self.b0.data_ptr(), h[0].data_ptr(), c[0].data_ptr(), stream))
        capi.check(lib.kvae_lstm128_step_fused(
            self.B, 0, h[0].data_ptr(), self.wih1t.data_ptr(), prev(1, 0), prev(1, 1),
            self.whh1t.data_ptr(), self.b1.data_ptr(), h[1].data_ptr(), c[1].data_ptr(), stream))
        capi.check(lib.kvae_softmax_head_forward(
            self.B, 128, self.K, h[1].data_ptr(), self.Wc.data_ptr(), self.bc.data_ptr(),
            out.data_ptr(), 0, stream))
        self.now = nxt

    @property
    def hidden(self):
        """(h_n, c_n), each (2,B,H), after the steps so far; None before the first."""
        return None if self.now is None else self.state[self.now]


def fused_lstm(x, lstm):
    """Run ``lstm`` (a torch.nn.LSTM, unidirectional, sequence-first) on x with
    zero initial state through the HIP recurrence.  Returns (output, (h_n, c_n))."""
    params = []
    for l in range(lstm.num_layers):
        params += [getattr(lstm, f"weight_ih_l{l}"), getattr(lstm, f"weight_hh_l{l}"),
                   getattr(lstm, f"bias_ih_l{l}"), getattr(lstm, f"bias_hh_l{l}")]
    two_layer_128 = lstm.num_layers == 2 and lstm.hidden_size == 128 and PERSISTENT
    if two_layer_128 and (x.shape[2] <= 4 or RESIDENT):
        fn = ResidentLSTM2Function if RESIDENT else FusedLSTM2Function
        out, h0, c0, c1 = fn.apply(x, *params)
        return out, LastState((h0[-1], out.detach()[-1]), (c0[-1], c1[-1]))
    out, h_n, c_n = FusedLSTMFunction.apply(x, *params)
    return out, (h_n, c_n)


def fused_lstm_head(x, lstm, linear, lead=False):
    """softmax(linear(lstm(x))) for the default network (two layers of 128 units, 2 to 4
    softmax outputs) as one autograd node whose backward runs the head's adjoint inside the
    upper layer's kernel; None when this form does not apply (use fused_lstm and
    softmax_head).  Returns (weights, state to continue from)."""
    K = linear.out_features
    if not (RESIDENT and PERSISTENT and lstm.num_layers == 2 and lstm.hidden_size == 128
            and not lstm.bidirectional and 2 <= K <= 4 and x.dtype == torch.float32
            and linear.bias is not None):
        return None
    params = []
    for l in range(2):
        params += [getattr(lstm, f"weight_ih_l{l}"), getattr(lstm, f"weight_hh_l{l}"),
                   getattr(lstm, f"bias_ih_l{l}"), getattr(lstm, f"bias_hh_l{l}")]
    alpha, h1, h0, c0, c1 = ResidentLSTM2HeadFunction.apply(x, *params, linear.weight, linear.bias,
                                                            lead)
    return alpha, LastState((h0[-1], h1[-1]), (c0[-1], c1[-1]))


class LastState:
    """(h_n, c_n) of a whole-sequence call, stacked over the layers only when
    something asks for it (a training step never does)."""

    def __init__(self, h_layers, c_layers):
        self._h, self._c, self._value = h_layers, c_layers, None

    def get(self):
        if self._value is None:
            self._value = (torch.stack([t.detach() for t in self._h]),
                           torch.stack([t.detach() for t in self._c]))
            self._h = self._c = None
        return self._value

    def __iter__(self):
        return iter(self.get())

    def __getitem__(self, i):
        return self.get()[i]
