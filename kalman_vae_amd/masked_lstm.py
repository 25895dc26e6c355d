"""The masked filter with the reference's default weight network (two LSTM layers of 128 units
and a softmax head) under autograd, as ONE autograd node.

With an observation mask the network is called once per time step on an observation that
depends on the filter's own prediction (kvae/state_space_model.py:219-352, the imputation of
:232-277 feeding kvae/dynamics_parameter_network.py:46-58), so neither direction is one launch;
but every per-step piece has a kernel of this library, and recording a few hundred small torch
operations per step (and walking them again backwards) is what the step-by-step path spends
its time on.  Here the forward pass is the stepping loop of the inference path with the
training kernels (which keep the activated gates), writing into slices of (T, ...) tensors,
and the backward pass walks the steps in reverse by hand:

  per step, last to first: the adjoint of the one-step filter segment (kvae_filter_backward
  with T = 1, the cotangents of its predicted state carried from the step after), then - the
  gradient w.r.t. the weights of step t+1 being complete - the head's and the two layers'
  adjoints of step t (kvae_softmax_head_backward, kvae_lstm128_cell_backward and three
  row-streaming products; the state gradients are carried here), then the adjoint of the
  imputation (kvae_impute_observation_backward), whose gradients w.r.t. the mixed matrices of
  step t ride into the segment of step t-1 as cotangents of its second row of matrices;
  after the loop: the LSTM's and the head's parameter gradients from the gate gradients of all
  steps at once, exactly as the unmasked network's backward pass forms them.
"""
import ctypes as C

import torch

from . import _capi as capi
from . import functions as F
from . import lstm_function as L

H = 128


def supported(a, lstm, linear, n, p, K):
    """Whether masked_filter_lstm_grad covers this network and these dimensions."""
    return (a.is_cuda and a.dtype == torch.float32 and isinstance(lstm, torch.nn.LSTM)
            and lstm.num_layers == 2 and lstm.hidden_size == H and lstm.input_size == p
            and 1 <= p <= 4 and not lstm.bidirectional and lstm.bias and lstm.proj_size == 0
            and linear.in_features == H and linear.out_features == K and 2 <= K <= 8
            and linear.bias is not None
            and F.embedding_dims(n, p, K, a.dtype) == (n, p, K)
            and F.impute_supported(n, p, K, a.dtype, a.device))


def _filter_step(dims, a_t, alpha2, mask_t, A_K, C_K, Q, R, m_in, P_in, m, P, mn, Pn, As2, Cs2):
    io = capi.fill(capi.FilterFwd(), a=a_t, alpha=alpha2, mask=mask_t, A_K=A_K, C_K=C_K, Q=Q, R=R,
                   m_in=m_in, P_in=P_in, carry_per_sequence=1, mean_filt=m, cov_filt=P,
                   mean_pred=mn, cov_pred=Pn, mat_As=As2, mat_Cs=Cs2,
                   status=F.status_word(a_t.device))
    capi.check(capi.load().kvae_filter_forward(C.byref(dims), C.byref(io),
                                               capi.current_stream_ptr()))


class MaskedFilterLSTMFunction(torch.autograd.Function):
    """(a, mask, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, w_head, b_head, A_K, C_K,
    Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R) ->
    (mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs, a_w, alpha (T+1,B,K),
    h_n (2,B,128), c_n (2,B,128): the network's state after the last step).
    m_in (n) / P_in (n,n): the state every sequence starts from; the network starts from the
    zero state.  eps_z (T,B,n) / eps_a (T,B,p): the imputation's draws or None."""

    @staticmethod
    def forward(ctx, a, mask, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, w_head,
                b_head, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R, symmetrize, burn_in,
                learn):
        lib = capi.load()
        capi.require_gpu(a, mask, A_K, C_K, Q, R, m_in, P_in)
        a, mask, A_K, C_K, Q, R, eps_z, L_Q, eps_a, L_R = map(
            F._c, (a, mask, A_K, C_K, Q, R, eps_z, L_Q, eps_a, L_R))
        T, B, p = a.shape
        K, n = A_K.shape[0], A_K.shape[1]
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
        As, Cs, aw = new(T + 1, B, n, n), new(T + 1, B, p, n), new(T, B, p)
        ws = new(T + 1, B, K)
        ws[0] = 1.0 / K
        gates0, h0, c0 = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
        gates1, h1, c1 = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
        with torch.no_grad():
            whh0t, wih1t, whh1t, b0, b1 = L._lstm2_operands(w_hh0, b_ih0, b_hh0, w_ih1, w_hh1,
                                                            b_ih1, b_hh1)
            wih0, wh, bh = w_ih0.contiguous(), w_head.contiguous(), b_head.contiguous()
            m0 = m_in.expand(B, n).contiguous()
            P0 = P_in.expand(B, n, n).contiguous()
        ctx.product_form = L._ProductForm.current()
        stream = capi.current_stream_ptr()
        step_dims = [F._dims(1, B, n, p, K, a.dtype, symmetrize, flag) for flag in (0, 1)]
        ptr = lambda t: None if t is None else t.data_ptr()
        m_pred, P_pred = m0, P0
        for t in range(T):
            F.impute_observation(ws[t], A_K, C_K, m_pred, a[t], mask[t],
                                 None if eps_z is None else eps_z[t], L_Q,
                                 None if eps_a is None else eps_a[t], L_R, out=aw[t])
            capi.check(lib.kvae_lstm128_forward(
                1, B, p, aw[t].data_ptr(), wih0.data_ptr(), b0.data_ptr(), whh0t.data_ptr(),
                gates0[t].data_ptr(), h0[t].data_ptr(), c0[t].data_ptr(),
                ptr(h0[t - 1] if t else None), ptr(c0[t - 1] if t else None), 1, stream))
            L._rows_matmul(h0[t], wih1t, b1, gates1[t])
            capi.check(lib.kvae_lstm128_forward(
                1, B, 0, None, None, None, whh1t.data_ptr(), gates1[t].data_ptr(),
                h1[t].data_ptr(), c1[t].data_ptr(), ptr(h1[t - 1] if t else None),
                ptr(c1[t - 1] if t else None), 1, stream))
            capi.check(lib.kvae_softmax_head_forward(B, H, K, h1[t].data_ptr(), wh.data_ptr(),
                                                     bh.data_ptr(), ws[t + 1].data_ptr(), 0, stream))
            _filter_step(step_dims[1 if t < burn_in else 0], a[t:t + 1], ws[t:t + 2],
                         mask[t:t + 1], A_K, C_K, Q, R, m_pred, P_pred, m[t:t + 1], P[t:t + 1],
                         mn[t:t + 1], Pn[t:t + 1], As[t:t + 2], Cs[t:t + 2])
            m_pred, P_pred = mn[t], Pn[t]
        ctx.save_for_backward(a, mask, w_ih0, w_hh0, w_ih1, w_hh1, wh, A_K, C_K, Q, R, m0, P0,
                              eps_z, L_Q, eps_a, L_R, m, P, mn, Pn, As, Cs, aw, ws, gates0, h0, c0,
                              gates1, h1, c1)
        ctx.flags = (bool(symmetrize), int(burn_in), bool(learn), m_in.dim(), P_in.dim())
        ctx.set_materialize_grads(False)
        # the state the network is left in, as after the equivalent per-step calls
        h_n, c_n = torch.stack((h0[-1], h1[-1])), torch.stack((c0[-1], c1[-1]))
        ctx.mark_non_differentiable(h_n, c_n)
        return m, P, mn, Pn, As, Cs, aw, ws, h_n, c_n

    @staticmethod
    def backward(ctx, g_m, g_P, g_mn, g_Pn, g_As, g_Cs, g_aw, g_ws, _g_h, _g_c):
        (a, mask, w_ih0, w_hh0, w_ih1, w_hh1, wh, A_K, C_K, Q, R, m0, P0, eps_z, L_Q, eps_a, L_R,
         m, P, mn, Pn, As, Cs, aw, ws, gates0, h0, c0, gates1, h1, c1) = ctx.saved_tensors
        symmetrize, burn_in, learn, m_dim, P_dim = ctx.flags
        lib = capi.load()
        T, B, p = a.shape
        K, n = A_K.shape[0], A_K.shape[1]
        dt, dev = a.dtype, a.device
        new = lambda *s: torch.empty(s, dtype=dt, device=dev)
        # (filled by a kernel: torch.zeros is a memset, whose replay order inside a captured
        # graph is not reliable on this ROCm version, see graphed.py)
        zeros = lambda *s: torch.full(s, 0.0, dtype=dt, device=dev)
        own = lambda g, *s: zeros(*s) if g is None else g.contiguous().clone()
        ro = lambda g, *s: zeros(*s) if g is None else g.contiguous()
        G_m, G_P = ro(g_m, T, B, n), ro(g_P, T, B, n, n)                  # read only
        G_mn, G_Pn = own(g_mn, T, B, n), own(g_Pn, T, B, n, n)            # carried cotangents join
        gA, gC = own(g_As, T + 1, B, n, n), own(g_Cs, T + 1, B, p, n)
        GW = own(g_ws, T + 1, B, K)
        D_a = new(T, B, p)
        dAK, dCK, dQ, dR = new(T, K, n, n), new(T, K, p, n), new(T, n, n), new(T, p, p)
        d_alpha2, d_m_in, d_P_in = new(2, B, K), new(B, n), new(B, n, n)
        dG0, dG1 = new(T, B, 4 * H), new(T, B, 4 * H)
        dc0, dc1 = [new(B, H), new(B, H)], [new(B, H), new(B, H)]
        dh0_rec, dh1_rec, dh0_up, dh1_head = new(B, H), new(B, H), new(B, H), new(B, H)
        through = new(B, p)
        head_part = new(lib.kvae_softmax_head_backward_blocks(B), K * (H + 1))
        DAI, DZ = zeros(T, B, p), zeros(T, B, n)
        wih0, whh0, wih1, whh1 = (w.contiguous() for w in (w_ih0, w_hh0, w_ih1, w_hh1))
        g_aw = None if g_aw is None else g_aw.contiguous()
        stream = capi.current_stream_ptr()
        step_dims = [F._dims(1, B, n, p, K, dt, symmetrize, flag) for flag in (0, 1)]
        work = [F._workspace(d, capi.OP_FILTER_BWD, dev) for d in step_dims]
        ptr = lambda t: None if t is None else t.data_ptr()
        with L._ProductForm(ctx.product_form):
            for t in range(T - 1, -1, -1):
                m_pred, P_pred = (mn[t - 1], Pn[t - 1]) if t else (m0, P0)
                which = 1 if t < burn_in else 0
                io = capi.FilterBwd()
                capi.fill(io.fwd, a=a[t:t + 1], alpha=ws[t:t + 2], mask=mask[t:t + 1], A_K=A_K,
                          C_K=C_K, Q=Q, R=R, m_in=m_pred, P_in=P_pred, carry_per_sequence=1,
                          mean_filt=m[t:t + 1], cov_filt=P[t:t + 1], mean_pred=mn[t:t + 1],
                          cov_pred=Pn[t:t + 1], mat_As=As[t:t + 2], mat_Cs=Cs[t:t + 2])
                capi.fill(io, g_mean_filt=G_m[t:t + 1], g_cov_filt=G_P[t:t + 1],
                          g_mean_pred=G_mn[t:t + 1], g_cov_pred=G_Pn[t:t + 1],
                          g_mat_As=gA[t:t + 2], g_mat_Cs=gC[t:t + 2], d_a=D_a[t:t + 1],
                          d_alpha=d_alpha2, d_m_in=d_m_in, d_P_in=d_P_in, d_A_K=dAK[t],
                          d_C_K=dCK[t], d_Q=dQ[t], d_R=dR[t])
                ws_buf, nbytes = work[which]
                capi.check(lib.kvae_filter_backward(C.byref(step_dims[which]), C.byref(io),
                                                    ws_buf.data_ptr(), nbytes, stream))
                GW[t:t + 2] += d_alpha2
                d_aw_a, d_aw_b = (None if g_aw is None else g_aw[t]), None
                if learn:
                    # the network's call of step t: a_w[t] -> alpha[t + 1]
                    last = t == T - 1
                    capi.check(lib.kvae_softmax_head_backward(
                        B, H, K, h1[t].data_ptr(), wh.data_ptr(), ws[t + 1].data_ptr(),
                        GW[t + 1].data_ptr(), dh1_head.data_ptr(), head_part.data_ptr(), stream))
                    capi.check(lib.kvae_lstm128_cell_backward(
                        B, gates1[t].data_ptr(), ptr(c1[t - 1] if t else None), c1[t].data_ptr(),
                        dh1_head.data_ptr(), None if last else dh1_rec.data_ptr(),
                        None if last else dc1[t & 1].data_ptr(), dG1[t].data_ptr(),
                        dc1[(t & 1) ^ 1].data_ptr(), None, 0, None, None, stream))
                    if t:
                        L._rows_matmul(dG1[t], whh1, None, dh1_rec)
                    L._rows_matmul(dG1[t], wih1, None, dh0_up)
                    # (the lower layer's kernel also forms the gradient w.r.t. its narrow input)
                    capi.check(lib.kvae_lstm128_cell_backward(
                        B, gates0[t].data_ptr(), ptr(c0[t - 1] if t else None), c0[t].data_ptr(),
                        dh0_up.data_ptr(), None if last else dh0_rec.data_ptr(),
                        None if last else dc0[t & 1].data_ptr(), dG0[t].data_ptr(),
                        dc0[(t & 1) ^ 1].data_ptr(), wih0.data_ptr(), p, None, through.data_ptr(),
                        stream))
                    if t:
                        L._rows_matmul(dG0[t], whh0, None, dh0_rec)
                    d_aw_a, d_aw_b = through, d_aw_a
                if d_aw_a is None:
                    # nothing arrives at the imputed observation (the weights are constants of
                    # the graph and a_w has no cotangent); the cotangents of this step's
                    # matrices are spent
                    gA[t].fill_(0.0)
                    gC[t].fill_(0.0)
                    if t:
                        G_mn[t - 1] += d_m_in
                        G_Pn[t - 1] += d_P_in
                    continue
                # the imputation a_w = mask a + (1 - mask) (C_t (A_t m_pred + L_Q eps_z) + L_R eps_a):
                # one launch.  It leaves the cotangents of the mixed matrices of step t in gA[t],
                # gC[t] (the external ones were consumed by this step's segment; these are
                # consumed by the segment of step t - 1 as its second row of matrices, and after
                # the loop for t = 0) and hands the segment's and its own gradient w.r.t. the
                # predicted state to step t - 1
                capi.check(lib.kvae_impute_observation_backward(
                    C.byref(step_dims[0]), d_aw_a.data_ptr(), ptr(d_aw_b), mask[t].data_ptr(),
                    As[t].data_ptr(), Cs[t].data_ptr(), m_pred.data_ptr(),
                    ptr(None if eps_z is None else eps_z[t]), ptr(L_Q), D_a[t].data_ptr(),
                    DAI[t].data_ptr(), DZ[t].data_ptr(), gC[t].data_ptr(), gA[t].data_ptr(),
                    d_m_in.data_ptr(), d_P_in.data_ptr(), ptr(G_mn[t - 1] if t else None),
                    ptr(G_Pn[t - 1] if t else None), stream))
            grads_net = (None,) * 10
            if learn:
                # parameters of the two layers from the gate gradients of all steps
                R_ = T * B
                d0, d1 = dG0.view(R_, 4 * H), dG1.view(R_, 4 * H)
                split = T > 1 and lib.kvae_lstm128_get_precision() == 1
                if split:
                    E = 4 * H * H
                    S = lib.kvae_lstm128_wgrad_slabs(R_)
                    slabs = new(S, 3 * E)
                    for i, (dg, hh, shift) in enumerate(((d0, h0, B), (d1, h0, 0), (d1, h1, B))):
                        capi.check(lib.kvae_lstm128_wgrad(R_, dg.data_ptr(), hh.data_ptr(), shift,
                                                          slabs[:, i * E:].data_ptr(), 3 * E, stream))
                    dw_hh0, dw_ih1, dw_hh1 = capi.column_sum(slabs).view(3, 4 * H, H).unbind(0)
                elif T > 1:
                    dw_hh0, dw_ih1, dw_hh1 = L._splitk_tn_many([
                        (d0[B:], h0[:-1].reshape(-1, H)), (d1, h0.reshape(R_, H)),
                        (d1[B:], h1[:-1].reshape(-1, H))])
                else:
                    dw_hh0, dw_hh1 = torch.full_like(w_hh0, 0.0), torch.full_like(w_hh1, 0.0)
                    dw_ih1 = d1.t() @ h0.view(R_, H)
                dw_ih0 = L._splitk_tn(d0, aw.view(R_, p))
                # (the library's column sum: torch's many-row reduction zeroes a semaphore with a
                # memset, which a captured graph does not replay reliably here)
                db0, db1 = capi.column_sum_tall(d0), capi.column_sum_tall(d1)
                # the head's parameters: its adjoint over all steps at once
                blocks = lib.kvae_softmax_head_backward_blocks(R_)
                part = new(blocks, K * (H + 1))
                scratch = new(R_, H)
                capi.check(lib.kvae_softmax_head_backward(
                    R_, H, K, h1.data_ptr(), wh.data_ptr(), ws[1:].data_ptr(),
                    GW[1:].contiguous().data_ptr(), scratch.data_ptr(), part.data_ptr(), stream))
                sh = capi.column_sum(part)
                grads_net = (dw_ih0, dw_hh0, db0, db0, dw_ih1, dw_hh1, db1, db1,
                             sh[:K * H].view(K, H), sh[K * H:])
        # what the imputation of step 0 left for the matrices A(alpha_0), C(alpha_0)
        # (sums of outer products as multiply + sum: these shapes make degenerate library GEMMs)
        w0 = ws[0].reshape(B, K, 1, 1)
        tall = lambda x, *shape: capi.column_sum_tall(x.reshape(x.shape[0], -1)).view(*shape)
        d_A_K = dAK.sum(0) + tall(w0 * gA[0].unsqueeze(1), K, n, n)
        d_C_K = dCK.sum(0) + tall(w0 * gC[0].unsqueeze(1), K, p, n)
        d_L_Q = d_L_R = None
        if eps_z is not None:
            d_L_Q = tall((DZ.unsqueeze(-1) * eps_z.unsqueeze(-2)).view(T * B, n * n), n, n)
        if eps_a is not None:
            d_L_R = tall((DAI.unsqueeze(-1) * eps_a.unsqueeze(-2)).view(T * B, p * p), p, p)
        d_m = d_m_in.sum(0) if m_dim == 1 else d_m_in
        d_P = d_P_in.sum(0) if P_dim == 2 else d_P_in
        return ((D_a, None) + grads_net + (d_A_K, d_C_K, dQ.sum(0), dR.sum(0), d_m, d_P, None,
                                           d_L_Q, None, d_L_R, None, None, None))


def masked_filter_lstm_grad(a, mask, lstm, linear, A_K, C_K, Q, R, m_in, P_in, eps_z=None, L_Q=None,
                            eps_a=None, L_R=None, symmetrize=True, burn_in=0,
                            learn_weight_model=True):
    """The masked filter with the network ``softmax(linear(lstm(.)))`` (torch.nn.LSTM with two
    layers of 128 units, started from the zero state, and a torch.nn.Linear) in the recursion,
    differentiable: see MaskedFilterLSTMFunction.  Returns (mean_filt, cov_filt, mean_pred,
    cov_pred, mat_As, mat_Cs, a_w, alpha (T+1,B,K), h_n, c_n), or None when the network or the
    dimensions are not covered."""
    K, n = A_K.shape[0], A_K.shape[1]
    if not supported(a, lstm, linear, n, a.shape[2], K):
        return None
    return MaskedFilterLSTMFunction.apply(
        a, mask, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
        lstm.weight_ih_l1, lstm.weight_hh_l1, lstm.bias_ih_l1, lstm.bias_hh_l1, linear.weight,
        linear.bias, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R, bool(symmetrize),
        int(burn_in), bool(learn_weight_model))
