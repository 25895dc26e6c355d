"""torch.autograd.Function wrappers over the HIP kernels (C ABI in
include/kvae_kalman.h).  PyTorch is used here for device memory, the current
stream and the autograd graph only; all arithmetic of the three ops and of
their backward passes runs in kalman_vae_amd/csrc/kernels.hip.

Each Function mirrors one piece of the reference:
  KalmanFilterFunction  - recursion of kvae/state_space_model.py:284-352
  KalmanSmoothFunction  - kvae/state_space_model.py:375-484
  KalmanElboFunction    - kvae/kalman_vae.py:191-242 (before `aggregate`)
"""
import ctypes as C

import math

import torch

from . import _capi as capi

# A persistent int32 status word per device.  Kernels OR failure bits into it;
# read_status() fetches and clears it (one host sync).
_status_words = {}


def status_word(device):
    key = (device.type, device.index)
    if key not in _status_words:
        _status_words[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _status_words[key]


def read_status(device, clear=True):
    word = status_word(device)
    value = int(word.item())
    if clear and value:
        word.zero_()
    return value


def raise_on_status(device):
    value = read_status(device)
    if value:
        raise ValueError("Kalman kernels: " + capi.status_message(value))


def _c(t):
    return None if t is None else t.contiguous()


def _dims(T, B, n, p, K, dtype, symmetrize, burn_in):
    lib = capi.load()
    code = capi.dtype_code(dtype)
    if not lib.kvae_supported(n, p, K, code):
        raise NotImplementedError(
            f"no HIP kernel compiled for z_dim={n}, a_dim={p}, K={K}; the public "
            "kalman_* functions of this module embed such a problem into a compiled one")
    return capi.Dims(T, B, n, p, K, code, int(bool(symmetrize)), int(max(0, burn_in)))


def _workspace(dims, op, device):
    nbytes = capi.load().kvae_workspace_bytes(C.byref(dims), op)
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=device), int(nbytes)


# ---------------------------------------------------------------------------
# Dimension embedding.  Kernels exist for a fixed list of (z_dim, a_dim, K).  Any
# other combination that fits inside a compiled one is solved EXACTLY as a block of
# the larger problem: with
#   A' = diag(A, 0), C' = [[C, 0], [0, 0]], Q' = diag(Q, I), R' = diag(R, I),
#   m0' = (m0, 0), P0' = diag(P0, I), a' = (a, 0), alpha' = (alpha, 0), eps' = (eps, 0)
# the extra state stays N(0, I), independent of the real block (every cross term is an
# exact zero), the filter / smoother outputs of the real block are unchanged, and each
# extra dimension adds the constant -log(2 pi)/2 to a Gaussian log-density term.  The
# padding and the slicing are differentiable torch ops, so gradients flow through.
# ---------------------------------------------------------------------------
LOG_2PI = 1.8378770664093453
MAX_Z_DIM, MAX_A_DIM, MAX_K = 32, 16, 8
_compiled = {}


def compiled_shapes(dtype, fused=False):
    """Every (z_dim, a_dim, K) the library has kernels for (``fused``: the one-launch
    objective), probed once."""
    code = capi.dtype_code(dtype)
    key = (code, bool(fused))
    if key not in _compiled:
        lib = capi.load()
        probe = lib.kvae_loss_supported if fused else lib.kvae_supported
        _compiled[key] = [(n, p, K) for n in range(1, MAX_Z_DIM + 1)
                          for p in range(1, MAX_A_DIM + 1) for K in range(1, MAX_K + 1)
                          if probe(n, p, K, code)]
    return _compiled[key]


def embedding_dims(n, p, K, dtype, fused=False):
    """The cheapest compiled (n2, p2, K2) with n2 >= n, p2 >= p, K2 >= K (K None: the
    op does not depend on K), or None."""
    fits = [s for s in compiled_shapes(dtype, fused)
            if s[0] >= n and s[1] >= p and (K is None or s[2] >= K)]
    return min(fits) if fits else None


def _require_embedding(n, p, K, dtype):
    dims = embedding_dims(n, p, K, dtype)
    if dims is None:
        raise NotImplementedError(
            f"no HIP kernel covers z_dim={n}, a_dim={p}, K={K} (limits: z_dim <= "
            f"{MAX_Z_DIM}, a_dim <= {MAX_A_DIM}, K <= {MAX_K}); add the combination to "
            "KVAE_DIM_LIST in kalman_vae_amd/csrc/launch_common.h")
    return dims


class _Embedding:
    """Pads operands of an (n, p, K) problem to (n2, p2, K2) and cuts results back."""

    def __init__(self, n, p, K, n2, p2, K2):
        self.n, self.p, self.K, self.n2, self.p2, self.K2 = n, p, K, n2, p2, K2
        self.dn, self.dp, self.dK = n2 - n, p2 - p, (K2 - K) if K is not None else 0
        self.identity = (self.dn, self.dp, self.dK) == (0, 0, 0)
        # what one (t, b) pair of each padded log-density term lacks
        self.c_obs, self.c_lat = 0.5 * self.dp * LOG_2PI, 0.5 * self.dn * LOG_2PI

    @staticmethod
    def _pad(t, *widths):
        if t is None or not any(widths):
            return t
        spec = []
        for w in widths:            # last dimension first
            spec += [0, w]
        return torch.nn.functional.pad(t, spec)

    @staticmethod
    def _tail(k, k2, like):
        key = ("tail", k, k2, like.dtype, like.device.type, like.device.index)
        if key not in _constant_vectors:
            d = torch.zeros(k2, dtype=like.dtype, device=like.device)
            d[k:] = 1
            _constant_vectors[key] = torch.diag(d)
        return _constant_vectors[key]

    def vec_n(self, t): return self._pad(t, self.dn)
    def vec_p(self, t): return self._pad(t, self.dp)
    def weights(self, t): return self._pad(t, self.dK)
    def mat_nn(self, t): return self._pad(t, self.dn, self.dn)
    def mat_pn(self, t): return self._pad(t, self.dn, self.dp)
    def bases_A(self, t): return self._pad(t, self.dn, self.dn, self.dK)
    def bases_C(self, t): return self._pad(t, self.dn, self.dp, self.dK)

    def cov_n(self, t):
        if t is None or not self.dn:
            return t
        return self._pad(t, self.dn, self.dn) + self._tail(self.n, self.n2, t)

    def cov_p(self, t):
        if t is None or not self.dp:
            return t
        return self._pad(t, self.dp, self.dp) + self._tail(self.p, self.p2, t)

    def cut_n(self, t): return t[..., :self.n].contiguous() if self.dn else t
    def cut_p(self, t): return t[..., :self.p].contiguous() if self.dp else t
    def cut_K(self, t): return t[..., :self.K].contiguous() if self.dK else t
    def cut_nn(self, t): return t[..., :self.n, :self.n].contiguous() if self.dn else t

    def cut_pn(self, t):
        return t[..., :self.p, :self.n].contiguous() if (self.dn or self.dp) else t


def _embedding(n, p, K, dtype, fused=False):
    dims = embedding_dims(n, p, K, dtype, fused) if fused else _require_embedding(n, p, K, dtype)
    return None if dims is None else _Embedding(n, p, K, *dims)


def _any_K(n, p, dtype=torch.float32):
    """The smoother and ELBO kernels do not depend on K, but kernels are keyed on
    (n, p, K): use any K compiled for this (n, p)."""
    for s in compiled_shapes(dtype):
        if s[0] == n and s[1] == p:
            return s[2]
    raise NotImplementedError(f"no HIP kernel compiled for z_dim={n}, a_dim={p}")


class SmallCholeskyFunction(torch.autograd.Function):
    """Lower Cholesky factor of one small SPD matrix (the factor
    MultivariateNormal builds for Q and R, kvae/state_space_model.py:242-247,
    261-266, 409-413).  No host sync: failure sets bit 8 of the status word."""

    @staticmethod
    def forward(ctx, A):
        capi.require_gpu(A)
        A = A.contiguous()
        L = torch.empty_like(A)
        capi.check(capi.load().kvae_cholesky_forward(
            A.shape[-1], capi.dtype_code(A.dtype), A.data_ptr(), L.data_ptr(),
            status_word(A.device).data_ptr(), capi.current_stream_ptr()))
        ctx.save_for_backward(L)
        return L

    @staticmethod
    def backward(ctx, gL):
        (L,) = ctx.saved_tensors
        gL = gL.contiguous()
        gA = torch.empty_like(L)
        capi.check(capi.load().kvae_cholesky_backward(
            L.shape[-1], capi.dtype_code(L.dtype), L.data_ptr(), gL.data_ptr(), gA.data_ptr(),
            capi.current_stream_ptr()))
        return gA


def noise_factors(Lq, Lr, q_reg, r_reg):
    """(Q, R, chol(R)) = (Lq Lq^T + q_reg I, Lr Lr^T + r_reg I, chol(R)) in one launch,
    values only (no autograd graph): kvae/state_space_model.py:112-146."""
    capi.require_gpu(Lq)
    Lq, Lr = Lq.detach().contiguous(), Lr.detach().contiguous()
    Q, R, LR = torch.empty_like(Lq), torch.empty_like(Lr), torch.empty_like(Lr)
    capi.check(capi.load().kvae_noise_factors(
        Lq.shape[0], Lr.shape[0], capi.dtype_code(Lq.dtype), Lq.data_ptr(), Lr.data_ptr(),
        float(q_reg), float(r_reg), Q.data_ptr(), R.data_ptr(), LR.data_ptr(),
        status_word(Lq.device).data_ptr(), capi.current_stream_ptr()))
    return Q, R, LR


def cholesky_factor(A):
    if A.dim() != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("cholesky_factor expects one square matrix")
    return SmallCholeskyFunction.apply(A)


class TotalSumFunction(torch.autograd.Function):
    """Sum of all elements of a GPU tensor as a 0-d tensor (deterministic tree
    sum on the HIP library; the gradient is the upstream scalar broadcast)."""

    @staticmethod
    def forward(ctx, v):
        capi.require_gpu(v)
        lib = capi.load()
        vc = v.contiguous()
        n = vc.numel()
        out = torch.empty(1, dtype=v.dtype, device=v.device)
        scratch = torch.empty(max(1, lib.kvae_vector_sum_scratch_elems(1, n)), dtype=v.dtype,
                              device=v.device)
        capi.check(lib.kvae_vector_sum(1, n, capi.dtype_code(v.dtype), vc.data_ptr(),
                                       scratch.data_ptr(), out.data_ptr(),
                                       capi.current_stream_ptr()))
        ctx.shape = v.shape
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape)


class TermTotalsFunction(torch.autograd.Function):
    """(obs, trans, post), each (T,B) -> their three totals as a (3,) tensor: the
    reduction inside ``aggregate`` (kvae/misc.py:21-41) for all three ELBO terms
    in one launch when the terms are adjacent in memory (as kalman_elbo_terms
    returns them)."""

    @staticmethod
    def forward(ctx, obs, trans, post):
        capi.require_gpu(obs, trans, post)
        lib = capi.load()
        n = obs.numel()
        step = n * obs.element_size()
        adjacent = (obs.is_contiguous() and trans.is_contiguous() and post.is_contiguous()
                    and trans.data_ptr() == obs.data_ptr() + step
                    and post.data_ptr() == obs.data_ptr() + 2 * step)
        src = obs if adjacent else torch.stack([obs, trans, post])
        out = torch.empty(3, dtype=obs.dtype, device=obs.device)
        scratch = torch.empty(max(1, lib.kvae_vector_sum_scratch_elems(3, n)), dtype=obs.dtype,
                              device=obs.device)
        capi.check(lib.kvae_vector_sum(3, n, capi.dtype_code(obs.dtype), src.data_ptr(),
                                       scratch.data_ptr(), out.data_ptr(),
                                       capi.current_stream_ptr()))
        ctx.shape = obs.shape
        return out

    @staticmethod
    def backward(ctx, g):
        full = g.view(3, *([1] * len(ctx.shape))).expand(3, *ctx.shape).contiguous()
        return full[0], full[1], full[2]


def term_totals(obs, trans, post):
    return TermTotalsFunction.apply(obs, trans, post)


_constant_vectors = {}


def constant_vector(values, like):
    """A small constant tensor kept on the device (no host-to-device copy per step)."""
    key = (tuple(float(v) for v in values), like.dtype, like.device.type, like.device.index)
    if key not in _constant_vectors:
        _constant_vectors[key] = torch.tensor(key[0], dtype=like.dtype, device=like.device)
    return _constant_vectors[key]


def total_sum(v):
    return TotalSumFunction.apply(v)


class KalmanFilterFunction(torch.autograd.Function):
    """(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in) ->
    (mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs).

    m_in/P_in of shape (n)/(n,n) start every sequence from the same state;
    (B,n)/(B,n,n) continue a recursion (segment mode)."""

    @staticmethod
    def forward(ctx, a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, symmetrize, burn_in, want_mats):
        capi.require_gpu(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in)
        a, alpha, mask, A_K, C_K, Q, R, m_in, P_in = map(
            _c, (a, alpha, mask, A_K, C_K, Q, R, m_in, P_in))
        T, B, p = a.shape
        K, n = A_K.shape[0], A_K.shape[1]
        if alpha.shape != (T + 1, B, K):
            raise ValueError(f"alpha must have shape {(T + 1, B, K)}, got {tuple(alpha.shape)}")
        dims = _dims(T, B, n, p, K, a.dtype, symmetrize, burn_in)
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
        As = new(T + 1, B, n, n) if want_mats else None
        Cs = new(T + 1, B, p, n) if want_mats else None
        io = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                       m_in=m_in, P_in=P_in, carry_per_sequence=int(m_in.dim() == 2),
                       mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs,
                       status=status_word(a.device))
        capi.check(capi.load().kvae_filter_forward(C.byref(dims), C.byref(io),
                                                   capi.current_stream_ptr()))
        # the mixed per-step matrices are outputs of the forward call; the large-state
        # register-tile backward kernel reads them back instead of re-mixing (null makes
        # the library fall back to its LDS kernel)
        ctx.save_for_backward(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, m, P, mn, Pn,
                              *((As, Cs) if want_mats else ()))
        ctx.dims = dims
        ctx.set_materialize_grads(False)
        if not want_mats:
            As, Cs = a.new_empty(0), a.new_empty(0)
            ctx.mark_non_differentiable(As, Cs)
        return m, P, mn, Pn, As, Cs

    @staticmethod
    def backward(ctx, g_m, g_P, g_mn, g_Pn, g_As, g_Cs):
        a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, m, P, mn, Pn = ctx.saved_tensors[:13]
        As, Cs = ctx.saved_tensors[13:] if len(ctx.saved_tensors) > 13 else (None, None)
        # contiguous copies must outlive the launch: keep them in locals
        g_m, g_P, g_mn, g_Pn, g_As, g_Cs = map(_c, (g_m, g_P, g_mn, g_Pn, g_As, g_Cs))
        dims = ctx.dims
        T, B, n, p, K = dims.T, dims.B, dims.n, dims.p, dims.K
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        d_a, d_alpha = new(T, B, p), new(T + 1, B, K)
        d_m_in, d_P_in = new(B, n), new(B, n, n)
        d_A_K, d_C_K, d_Q, d_R = new(K, n, n), new(K, p, n), new(n, n), new(p, p)
        io = capi.FilterBwd()
        capi.fill(io.fwd, a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R, m_in=m_in,
                  P_in=P_in, carry_per_sequence=int(m_in.dim() == 2), mean_filt=m, cov_filt=P,
                  mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs)
        capi.fill(io, g_mean_filt=g_m, g_cov_filt=g_P, g_mean_pred=g_mn,
                  g_cov_pred=g_Pn, g_mat_As=g_As, g_mat_Cs=g_Cs, d_a=d_a,
                  d_alpha=d_alpha, d_m_in=d_m_in, d_P_in=d_P_in, d_A_K=d_A_K, d_C_K=d_C_K,
                  d_Q=d_Q, d_R=d_R)
        ws, nbytes = _workspace(dims, capi.OP_FILTER_BWD, a.device)
        capi.check(capi.load().kvae_filter_backward(C.byref(dims), C.byref(io), ws.data_ptr(),
                                                    nbytes, capi.current_stream_ptr()))
        if m_in.dim() == 1:
            d_m_in = d_m_in.sum(0) if ctx.needs_input_grad[7] else None
            d_P_in = d_P_in.sum(0) if ctx.needs_input_grad[8] else None
        return d_a, d_alpha, None, d_A_K, d_C_K, d_Q, d_R, d_m_in, d_P_in, None, None, None


class KalmanSmoothFunction(torch.autograd.Function):
    """(mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs, LR, eps_state,
    eps_obs) -> (mean_smooth, cov_smooth, zs, as_resampled)."""

    @staticmethod
    def forward(ctx, m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs, symmetrize, burn_in):
        capi.require_gpu(m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs)
        m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs = map(
            _c, (m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs))
        T, B, n = m.shape
        p = Cs.shape[2]
        dims = _dims(T, B, n, p, _any_K(n, p, m.dtype), m.dtype, symmetrize, burn_in)
        new = lambda *s: torch.empty(s, dtype=m.dtype, device=m.device)
        ms, Ps, zs, ares = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, p)
        io = capi.fill(capi.SmoothFwd(), mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn,
                       mat_As=As, mat_Cs=Cs, LR=LR, eps_state=eps_state, eps_obs=eps_obs,
                       mean_smooth=ms, cov_smooth=Ps, zs=zs, as_resampled=ares,
                       status=status_word(m.device))
        capi.check(capi.load().kvae_smooth_forward(C.byref(dims), C.byref(io),
                                                   capi.current_stream_ptr()))
        ctx.save_for_backward(m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs, ms, Ps)
        ctx.dims = dims
        ctx.set_materialize_grads(False)
        return ms, Ps, zs, ares

    @staticmethod
    def backward(ctx, g_ms, g_Ps, g_zs, g_ares):
        m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs, ms, Ps = ctx.saved_tensors
        g_ms, g_Ps, g_zs, g_ares = map(_c, (g_ms, g_Ps, g_zs, g_ares))  # kept alive past the launch
        dims = ctx.dims
        T, B, n, p = dims.T, dims.B, dims.n, dims.p
        new = lambda *s: torch.empty(s, dtype=m.dtype, device=m.device)
        d_m, d_P, d_mn, d_Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
        d_As, d_Cs = new(T + 1, B, n, n), new(T + 1, B, p, n)
        want_LR = LR is not None and eps_obs is not None and ctx.needs_input_grad[6]
        d_LR = new(p, p) if want_LR else None
        io = capi.SmoothBwd()
        capi.fill(io.fwd, mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn, mat_As=As,
                  mat_Cs=Cs, LR=LR, eps_state=eps_state, eps_obs=eps_obs, mean_smooth=ms,
                  cov_smooth=Ps)
        capi.fill(io, g_mean_smooth=g_ms, g_cov_smooth=g_Ps, g_zs=g_zs,
                  g_as_resampled=g_ares, d_mean_filt=d_m, d_cov_filt=d_P, d_mean_pred=d_mn,
                  d_cov_pred=d_Pn, d_mat_As=d_As, d_mat_Cs=d_Cs, d_LR=d_LR)
        ws, nbytes = _workspace(dims, capi.OP_SMOOTH_BWD, m.device)
        capi.check(capi.load().kvae_smooth_backward(C.byref(dims), C.byref(io), ws.data_ptr(),
                                                    nbytes, capi.current_stream_ptr()))
        return d_m, d_P, d_mn, d_Pn, d_As, d_Cs, d_LR, None, None, None, None


class KalmanElboFunction(torch.autograd.Function):
    """(a, mean_smooth, cov_smooth, mat_As, mat_Cs, Q, R, m0, P0, eps) ->
    (obs, trans, post) each (T,B), and the sample z (T,B,n)."""

    @staticmethod
    def forward(ctx, a, ms, Ps, As, Cs, Q, R, m0, P0, eps):
        capi.require_gpu(a, ms, Ps, As, Cs, Q, R, m0, P0, eps)
        a, ms, Ps, As, Cs, Q, R, m0, P0, eps = map(_c, (a, ms, Ps, As, Cs, Q, R, m0, P0, eps))
        T, B, n = ms.shape
        p = a.shape[2]
        dims = _dims(T, B, n, p, _any_K(n, p, a.dtype), a.dtype, True, 0)
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        # the three terms live in one (3,T,B) buffer so that they can be totalled by
        # one launch (term_totals)
        terms, z = new(3, T, B), new(T, B, n)
        obs, trans, post = terms[0], terms[1], terms[2]
        io = capi.fill(capi.ElboFwd(), a=a, mean_smooth=ms, cov_smooth=Ps, mat_As=As, mat_Cs=Cs,
                       Q=Q, R=R, m0=m0, P0=P0, eps=eps, obs=obs, trans=trans, post=post, z=z,
                       status=status_word(a.device))
        capi.check(capi.load().kvae_elbo_forward(C.byref(dims), C.byref(io),
                                                 capi.current_stream_ptr()))
        ctx.save_for_backward(a, ms, Ps, As, Cs, Q, R, m0, P0, eps)
        ctx.dims = dims
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(z)
        return obs, trans, post, z

    @staticmethod
    def backward(ctx, g_obs, g_trans, g_post, _g_z):
        a, ms, Ps, As, Cs, Q, R, m0, P0, eps = ctx.saved_tensors
        g_obs, g_trans, g_post = map(_c, (g_obs, g_trans, g_post))  # kept alive past the launch
        dims = ctx.dims
        T, B, n, p = dims.T, dims.B, dims.n, dims.p
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        d_a, d_ms, d_Ps = new(T, B, p), new(T, B, n), new(T, B, n, n)
        d_As, d_Cs = new(T + 1, B, n, n), new(T + 1, B, p, n)
        d_Q, d_R, d_m0, d_P0 = new(n, n), new(p, p), new(n), new(n, n)
        io = capi.ElboBwd()
        capi.fill(io.fwd, a=a, mean_smooth=ms, cov_smooth=Ps, mat_As=As, mat_Cs=Cs, Q=Q, R=R,
                  m0=m0, P0=P0, eps=eps)
        capi.fill(io, g_obs=g_obs, g_trans=g_trans, g_post=g_post, d_a=d_a,
                  d_mean_smooth=d_ms, d_cov_smooth=d_Ps, d_mat_As=d_As, d_mat_Cs=d_Cs, d_Q=d_Q,
                  d_R=d_R, d_m0=d_m0, d_P0=d_P0)
        ws, nbytes = _workspace(dims, capi.OP_ELBO_BWD, a.device)
        capi.check(capi.load().kvae_elbo_backward(C.byref(dims), C.byref(io), ws.data_ptr(),
                                                  nbytes, capi.current_stream_ptr()))
        return d_a, d_ms, d_Ps, d_As, d_Cs, d_Q, d_R, d_m0, d_P0, None


class KalmanElboObjectiveFunction(torch.autograd.Function):
    """The three ELBO terms AND their aggregation in one autograd node:
    (a, mean_smooth, cov_smooth, mat_As, mat_Cs, Q, R, m0, P0, eps; w_obs, w_trans,
    w_post) -> objective = w_obs sum(obs) + w_trans sum(trans) + w_post sum(post)
    (kvae/kalman_vae.py:191-260 with aggregate()'s constant folded into the weights),
    the three weighted totals (3,), the un-aggregated terms (3,T,B) and the sample z
    (both values only).  The backward pass hands the kernel one scalar per term instead
    of three (T,B) cotangent tensors, so neither pass runs any elementwise glue."""

    @staticmethod
    def forward(ctx, a, ms, Ps, As, Cs, Q, R, m0, P0, eps, w_obs, w_trans, w_post):
        capi.require_gpu(a, ms, Ps, As, Cs, Q, R, m0, P0, eps)
        a, ms, Ps, As, Cs, Q, R, m0, P0, eps = map(_c, (a, ms, Ps, As, Cs, Q, R, m0, P0, eps))
        lib = capi.load()
        T, B, n = ms.shape
        p = a.shape[2]
        dims = _dims(T, B, n, p, _any_K(n, p, a.dtype), a.dtype, True, 0)
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        terms, z, out = new(3, T, B), new(T, B, n), new(4)
        io = capi.fill(capi.ElboFwd(), a=a, mean_smooth=ms, cov_smooth=Ps, mat_As=As, mat_Cs=Cs,
                       Q=Q, R=R, m0=m0, P0=P0, eps=eps, obs=terms[0], trans=terms[1],
                       post=terms[2], z=z, status=status_word(a.device))
        stream = capi.current_stream_ptr()
        capi.check(lib.kvae_elbo_forward(C.byref(dims), C.byref(io), stream))
        scratch = new(lib.kvae_weighted_terms_scratch_elems(T * B))
        capi.check(lib.kvae_weighted_terms(T * B, capi.dtype_code(a.dtype), terms.data_ptr(),
                                           scratch.data_ptr(), w_obs, w_trans, w_post,
                                           out.data_ptr(), stream))
        ctx.save_for_backward(a, ms, Ps, As, Cs, Q, R, m0, P0, eps)
        ctx.dims, ctx.weights = dims, (float(w_obs), float(w_trans), float(w_post))
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(terms, z)
        return out[3], out[:3], terms, z

    @staticmethod
    def backward(ctx, g_objective, g_weighted, _g_terms, _g_z):
        if g_objective is None and g_weighted is None:
            return (None,) * 13
        a, ms, Ps, As, Cs, Q, R, m0, P0, eps = ctx.saved_tensors
        if g_weighted is None:
            scale, stride = g_objective.contiguous(), 0
        else:
            scale = g_weighted if g_objective is None else g_weighted + g_objective
            scale, stride = scale.contiguous(), 1
        dims = ctx.dims
        T, B, n, p = dims.T, dims.B, dims.n, dims.p
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        d_a, d_ms, d_Ps = new(T, B, p), new(T, B, n), new(T, B, n, n)
        d_As, d_Cs = new(T + 1, B, n, n), new(T + 1, B, p, n)
        d_Q, d_R, d_m0, d_P0 = new(n, n), new(p, p), new(n), new(n, n)
        io = capi.ElboBwd()
        capi.fill(io.fwd, a=a, mean_smooth=ms, cov_smooth=Ps, mat_As=As, mat_Cs=Cs, Q=Q, R=R,
                  m0=m0, P0=P0, eps=eps)
        capi.fill(io, g_scale=scale, g_scale_stride=stride, d_a=d_a, d_mean_smooth=d_ms,
                  d_cov_smooth=d_Ps, d_mat_As=d_As, d_mat_Cs=d_Cs, d_Q=d_Q, d_R=d_R, d_m0=d_m0,
                  d_P0=d_P0)
        io.w_obs, io.w_trans, io.w_post = ctx.weights
        ws, nbytes = _workspace(dims, capi.OP_ELBO_BWD, a.device)
        capi.check(capi.load().kvae_elbo_backward(C.byref(dims), C.byref(io), ws.data_ptr(),
                                                  nbytes, capi.current_stream_ptr()))
        return d_a, d_ms, d_Ps, d_As, d_Cs, d_Q, d_R, d_m0, d_P0, None, None, None, None


class KalmanLossFunction(torch.autograd.Function):
    """Filter -> smoother -> ELBO terms -> weighted totals as ONE autograd node (the sequence
    of kvae/kalman_vae.py:141-260 without an observation mask feeding the weight network):
    (a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps; w_obs, w_trans, w_post) ->
    (objective, weighted totals (3,), terms (3,T,B) or an empty tensor).
    None of the twelve filter / smoother tensors is materialised.

    When a gradient will be asked for, the forward pass already computes it
    (kvae_kalman_loss_value_and_grad: for a modest batch ONE launch whose per-step state never
    leaves the chip) for a unit cotangent of the objective; the gradient is linear in that
    cotangent, so the backward pass only multiplies.  A caller that differentiates the three
    weighted totals separately gets kvae_kalman_loss_backward with its cotangents instead."""

    @staticmethod
    def forward(ctx, a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, w_obs, w_trans, w_post,
                symmetrize, burn_in, want_terms):
        capi.require_gpu(a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps)
        a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps = map(
            _c, (a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps))
        lib = capi.load()
        T, B, p = a.shape
        K, n = A_K.shape[0], A_K.shape[1]
        if alpha.shape != (T + 1, B, K):
            raise ValueError(f"alpha must have shape {(T + 1, B, K)}, got {tuple(alpha.shape)}")
        if not lib.kvae_loss_supported(n, p, K, capi.dtype_code(a.dtype)):
            raise NotImplementedError(
                f"no fused-objective kernel compiled for z_dim={n}, a_dim={p}, K={K}")
        dims = capi.Dims(T, B, n, p, K, capi.dtype_code(a.dtype), int(bool(symmetrize)),
                         int(max(0, burn_in)))
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        saved = torch.empty(lib.kvae_loss_saved_bytes(C.byref(dims)), dtype=torch.uint8,
                            device=a.device)
        terms = new(3, T, B) if want_terms else None
        out = new(4)
        need = ctx.needs_input_grad
        with_grad = any(need[:10])
        ctx.dims, ctx.weights = dims, (float(w_obs), float(w_trans), float(w_post))
        if with_grad:
            # all gradients in ONE buffer (each 256-byte aligned): the backward pass scales
            # them by the cotangent with a single launch
            shapes = dict(d_a=(T, B, p), d_alpha=(T + 1, B, K), d_A_K=(K, n, n), d_C_K=(K, p, n),
                          d_Q=(n, n), d_R=(p, p), d_m0=(n,) if need[7] else None,
                          d_P0=(n, n) if need[8] else None)
            offsets, total = {}, 0
            for name, shape in shapes.items():
                if shape is not None:
                    offsets[name] = total
                    total += -(-math.prod(shape) // 64) * 64
            flat = new(total)
            grads = {name: None if shape is None else
                     flat[offsets[name]:offsets[name] + math.prod(shape)].view(shape)
                     for name, shape in shapes.items()}
            io = capi.LossBwd()
            capi.fill(io.fwd, a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R, m0=m0,
                      P0=P0, eps=eps, saved=saved, terms=terms, out=out,
                      status=status_word(a.device))
            io.fwd.w_obs, io.fwd.w_trans, io.fwd.w_post = ctx.weights
            unit = _unit_cotangent(a)
            capi.fill(io, g_scale=unit, g_scale_stride=0, **grads)
            ws, nbytes = _workspace(dims, capi.OP_LOSS_VALUE_AND_GRAD, a.device)
            capi.check(lib.kvae_kalman_loss_value_and_grad(C.byref(dims), C.byref(io),
                                                           ws.data_ptr(), nbytes,
                                                           capi.current_stream_ptr()))
            ctx.unit_grads, ctx.unit_flat = grads, flat
        else:
            io = capi.fill(capi.LossFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q,
                           R=R, m0=m0, P0=P0, eps=eps, saved=saved, terms=terms, out=out,
                           status=status_word(a.device))
            io.w_obs, io.w_trans, io.w_post = ctx.weights
            ws, nbytes = _workspace(dims, capi.OP_LOSS_FWD, a.device)
            capi.check(lib.kvae_kalman_loss_forward(C.byref(dims), C.byref(io), ws.data_ptr(),
                                                    nbytes, capi.current_stream_ptr()))
            ctx.unit_grads = None
        ctx.save_for_backward(a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, saved)
        ctx.set_materialize_grads(False)
        if terms is None:
            terms = a.new_empty(0)
        ctx.mark_non_differentiable(terms)
        return out[3], out[:3], terms

    @staticmethod
    def backward(ctx, g_objective, g_weighted, _g_terms):
        if g_objective is None and g_weighted is None:
            return (None,) * 16
        a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, saved = ctx.saved_tensors
        if g_weighted is None and ctx.unit_grads is not None:
            # the usual case: only the objective was differentiated
            g = ctx.unit_grads
            ctx.unit_grads = None
            ctx.unit_flat.mul_(g_objective.reshape(()))      # (padding between them included)
            ctx.unit_flat = None
            return (g["d_a"], g["d_alpha"], None, g["d_A_K"], g["d_C_K"], g["d_Q"], g["d_R"],
                    g["d_m0"], g["d_P0"], None, None, None, None, None, None, None)
        if g_weighted is None:
            scale, stride = g_objective.contiguous(), 0
        else:
            scale = g_weighted if g_objective is None else g_weighted + g_objective
            scale, stride = scale.contiguous(), 1
        dims = ctx.dims
        T, B, n, p, K = dims.T, dims.B, dims.n, dims.p, dims.K
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        d_a, d_alpha = new(T, B, p), new(T + 1, B, K)
        d_A_K, d_C_K, d_Q, d_R = new(K, n, n), new(K, p, n), new(n, n), new(p, p)
        need = ctx.needs_input_grad
        d_m0 = new(n) if need[7] else None
        d_P0 = new(n, n) if need[8] else None
        io = capi.LossBwd()
        capi.fill(io.fwd, a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R, m0=m0, P0=P0,
                  eps=eps, saved=saved)
        io.fwd.w_obs, io.fwd.w_trans, io.fwd.w_post = ctx.weights
        capi.fill(io, g_scale=scale, g_scale_stride=stride, d_a=d_a, d_alpha=d_alpha,
                  d_A_K=d_A_K, d_C_K=d_C_K, d_Q=d_Q, d_R=d_R, d_m0=d_m0, d_P0=d_P0)
        ws, nbytes = _workspace(dims, capi.OP_LOSS_BWD, a.device)
        capi.check(capi.load().kvae_kalman_loss_backward(C.byref(dims), C.byref(io), ws.data_ptr(),
                                                         nbytes, capi.current_stream_ptr()))
        return (d_a, d_alpha, None, d_A_K, d_C_K, d_Q, d_R, d_m0, d_P0, None, None, None, None,
                None, None, None)


_unit = {}


def _unit_cotangent(like):
    """A device scalar 1 of ``like``'s dtype (cached per device: no fill launch per step)."""
    key = (like.device, like.dtype)
    t = _unit.get(key)
    if t is None:
        t = _unit[key] = torch.ones(1, dtype=like.dtype, device=like.device)
    return t


def _shift_totals(objective, weighted, terms, count, shifts, like):
    """Add per-(t, b) constants ``shifts`` (already multiplied by the term weights for the
    totals) to the outputs of an objective call."""
    (c_obs, c_trans, c_post), (w_obs, w_trans, w_post) = shifts
    add = (w_obs * c_obs * count, w_trans * c_trans * count, w_post * c_post * count)
    weighted = weighted + constant_vector(add, like)
    objective = objective + sum(add)
    if terms is not None and terms.numel():
        terms = terms + constant_vector((c_obs, c_trans, c_post), like).view(3, 1, 1)
    return objective, weighted, terms


def kalman_loss(a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, w_obs, w_trans, w_post,
                symmetrize=True, burn_in=0, want_terms=False):
    """(objective, weighted totals (3,), terms (3,T,B) or empty): see KalmanLossFunction."""
    T, B, p = a.shape
    K, n = A_K.shape[0], A_K.shape[1]
    e = _embedding(n, p, K, a.dtype, fused=True)
    if e is None:
        raise NotImplementedError(
            f"no fused-objective kernel covers z_dim={n}, a_dim={p}, K={K}")
    w = (float(w_obs), float(w_trans), float(w_post))
    if e.identity:
        return KalmanLossFunction.apply(a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, *w,
                                        bool(symmetrize), int(burn_in), bool(want_terms))
    objective, weighted, terms = KalmanLossFunction.apply(
        e.vec_p(a), e.weights(alpha), mask, e.bases_A(A_K), e.bases_C(C_K), e.cov_n(Q),
        e.cov_p(R), e.vec_n(m0), e.cov_n(P0), e.vec_n(eps), *w, bool(symmetrize), int(burn_in),
        bool(want_terms))
    return _shift_totals(objective, weighted, terms, T * B, ((e.c_obs, e.c_lat, e.c_lat), w), a)


def loss_supported(n, p, K, dtype):
    """Whether the fused objective has a kernel for these dimensions (directly or as a
    block of a larger compiled problem)."""
    return embedding_dims(n, p, K, dtype, fused=True) is not None


@torch.no_grad()
def impute_observation(w, A_K, C_K, m_pred, a, mask, eps_z=None, L_Q=None, eps_a=None, L_R=None,
                       out=None):
    """One step of the masked filter's imputation for every sequence in one launch
    (kvae/state_space_model.py:232-277): returns a_w (B,p) = mask a + (1 - mask) C(w) z,
    z = A(w) m_pred (+ L_Q eps_z) (+ L_R eps_a on the observation), or None when no
    kernel is compiled for these dimensions (the caller then uses torch operations).
    ``out``: a contiguous (B,p) tensor to write a_w into (a slice of the caller's
    (T,B,p) result).  Values only: the step-by-step filter calls it when no gradient is
    recorded."""
    capi.require_gpu(w, A_K, C_K, m_pred, a, mask)
    B, K = w.shape
    n, p = m_pred.shape[1], a.shape[1]
    code = capi.dtype_code(a.dtype)
    dims2 = embedding_dims(n, p, K, a.dtype)
    if dims2 is not None and dims2 != (n, p, K):
        e = _Embedding(n, p, K, *dims2)
        a_w = impute_observation(e.weights(w), e.bases_A(A_K), e.bases_C(C_K), e.vec_n(m_pred),
                                 e.vec_p(a), mask, e.vec_n(eps_z), e.cov_n(L_Q), e.vec_p(eps_a),
                                 e.cov_p(L_R))
        if a_w is None:
            return None
        return e.cut_p(a_w) if out is None else out.copy_(e.cut_p(a_w))
    dims = capi.Dims(1, B, n, p, K, code, 1, 0)
    tensors = [_c(t) for t in (w, A_K, C_K, m_pred, a, mask, eps_z, L_Q, eps_a, L_R)]
    if out is not None and not (out.is_contiguous() and out.shape == (B, p) and out.dtype == a.dtype):
        raise ValueError("impute_observation: out must be a contiguous (B, a_dim) tensor of a's dtype")
    a_w = torch.empty(B, p, dtype=a.dtype, device=a.device) if out is None else out
    rc = capi.load().kvae_impute_observation(C.byref(dims), *(capi.ptr(t) for t in tensors),
                                             a_w.data_ptr(), capi.current_stream_ptr())
    if rc == -2:      # KVAE_E_UNSUPPORTED
        return None
    capi.check(rc)
    return a_w


@torch.no_grad()
def filter_step_into(a_t, alpha2, mask_t, A_K, C_K, Q, R, m_in, P_in, m, P, mn, Pn, As2, Cs2,
                     symmetrize=True):
    """One step of the filter (kvae_filter_forward with T = 1, per-sequence carry) that
    writes into the caller's tensors: a_t (1,B,p), alpha2 (2,B,K) = the weights of this
    step and the next, mask_t (1,B) or None, m_in (B,n), P_in (B,n,n); results m, mn
    (1,B,n), P, Pn (1,B,n,n), As2 (2,B,n,n), Cs2 (2,B,p,n) - contiguous slices of (T, ...)
    buffers, so a step-by-step caller never stacks per-step tensors.  Values only.  The
    dimensions must have a kernel of their own (``embedding_dims(n, p, K) == (n, p, K)``)."""
    tensors = (a_t, alpha2, mask_t, A_K, C_K, Q, R, m_in, P_in, m, P, mn, Pn, As2, Cs2)
    capi.require_gpu(*tensors)
    if not all(t is None or t.is_contiguous() for t in tensors):
        raise ValueError("filter_step_into: every operand must be contiguous")
    B, p = a_t.shape[1:]
    K, n = A_K.shape[0], A_K.shape[1]
    if (alpha2.shape != (2, B, K) or m_in.shape != (B, n) or P_in.shape != (B, n, n)
            or m.shape != (1, B, n) or P.shape != (1, B, n, n) or mn.shape != (1, B, n)
            or Pn.shape != (1, B, n, n) or As2.shape != (2, B, n, n) or Cs2.shape != (2, B, p, n)):
        raise ValueError("filter_step_into: operand shapes do not match (B, n, p, K)")
    dims = _dims(1, B, n, p, K, a_t.dtype, symmetrize, 0)
    io = capi.fill(capi.FilterFwd(), a=a_t, alpha=alpha2, mask=mask_t, A_K=A_K, C_K=C_K, Q=Q, R=R,
                   m_in=m_in, P_in=P_in, carry_per_sequence=1, mean_filt=m, cov_filt=P,
                   mean_pred=mn, cov_pred=Pn, mat_As=As2, mat_Cs=Cs2,
                   status=status_word(a_t.device))
    capi.check(capi.load().kvae_filter_forward(C.byref(dims), C.byref(io),
                                               capi.current_stream_ptr()))


_impute_kernel_exists = {}


def impute_supported(n, p, K, dtype, device):
    """Whether kvae_impute_observation has a kernel for these dimensions (probed once
    with a one-sequence call)."""
    key = (n, p, K, dtype)
    if key not in _impute_kernel_exists:
        z = lambda *s: torch.zeros(*s, dtype=dtype, device=device)
        _impute_kernel_exists[key] = impute_observation(
            z(1, K), z(K, n, n), z(K, p, n), z(1, n), z(1, p), z(1)) is not None
    return _impute_kernel_exists[key]


def pack_mlp2(first, last):
    """The two Linear layers of softmax(W2 relu(W1 a + b1) + b2) in the layout
    kvae_filter_forward_masked_mlp reads: per hidden unit j the row W1[j], b1[j] and the
    column W2[:, j]; then b2."""
    units = torch.cat([first.weight, first.bias.unsqueeze(1), last.weight.t()], dim=1)
    return torch.cat([units.reshape(-1), last.bias]).detach().contiguous()


def _pad_packed_mlp(e, packed, hidden):
    """The packed perceptron of an (n, p, K) problem for its embedding ``e``: zero columns
    for the extra inputs, and extra outputs whose softmax weight is exactly zero."""
    p, K = e.p, e.K
    units = packed[:hidden * (p + 1 + K)].view(hidden, p + 1 + K)
    units = torch.cat([e.vec_p(units[:, :p]), units[:, p:p + 1], e.weights(units[:, p + 1:])],
                      dim=1)
    off = packed.new_full((e.dK,), -1e30)
    return torch.cat([units.reshape(-1), packed[hidden * (p + 1 + K):], off]).contiguous()


def masked_mlp_fits(hidden, p, K, dtype):
    """Whether the packed two-layer network is within the 64 KB the kernels accept."""
    return (hidden * (p + 1 + K) + K) * torch.empty((), dtype=dtype).element_size() <= 64 * 1024


@torch.no_grad()
def masked_filter_mlp(a, mask, packed, hidden, A_K, C_K, Q, R, m_in, P_in, eps_z=None, L_Q=None,
                      eps_a=None, L_R=None, symmetrize=True, first_weights=None):
    """The masked filter (kvae/state_space_model.py:219-352) with a two-layer perceptron
    weight network evaluated inside the recursion: one launch for all T steps, values
    only.  Returns (mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs, a_w,
    weights (T+1,B,K)), or None when no kernel is compiled for these dimensions."""
    capi.require_gpu(a, mask, packed, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R)
    a, mask, packed, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R = map(
        _c, (a, mask, packed, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R))
    T, B, p = a.shape
    K, n = A_K.shape[0], A_K.shape[1]
    dims2 = embedding_dims(n, p, K, a.dtype)
    if dims2 is None:
        return None
    if dims2 != (n, p, K):
        # the same recursion on the padded problem: zero columns for the extra inputs of
        # the network, and extra outputs whose softmax weight is exactly zero
        e = _Embedding(n, p, K, *dims2)
        packed2 = _pad_packed_mlp(e, packed, hidden)
        if not masked_mlp_fits(hidden, e.p2, e.K2, a.dtype):
            return None
        out = masked_filter_mlp(e.vec_p(a), mask, packed2, hidden, e.bases_A(A_K),
                                e.bases_C(C_K), e.cov_n(Q), e.cov_p(R), e.vec_n(m_in),
                                e.cov_n(P_in), e.vec_n(eps_z), e.cov_n(L_Q), e.vec_p(eps_a),
                                e.cov_p(L_R), symmetrize=symmetrize, first_weights=K)
        if out is None:
            return None
        m, P, mn, Pn, As, Cs, a_w, alpha = out
        return (e.cut_n(m), e.cut_nn(P), e.cut_n(mn), e.cut_nn(Pn), e.cut_nn(As), e.cut_pn(Cs),
                e.cut_p(a_w), e.cut_K(alpha))
    dims = _dims(T, B, n, p, K, a.dtype, symmetrize, 0)
    new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
    m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
    As, Cs, a_w = new(T + 1, B, n, n), new(T + 1, B, p, n), new(T, B, p)
    alpha = new(T + 1, B, K)
    # the first step mixes with uniform weights over the real bases
    first = first_weights or K
    alpha[0] = 0.0
    alpha[0, :, :first] = 1.0 / first
    io = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                   m_in=m_in, P_in=P_in, carry_per_sequence=int(m_in.dim() == 2),
                   mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs,
                   status=status_word(a.device))
    net = capi.fill(capi.MaskedMlp(), hidden=int(hidden), packed=packed, eps_z=eps_z, L_Q=L_Q,
                    eps_a=eps_a, L_R=L_R, a_w=a_w)
    rc = capi.load().kvae_filter_forward_masked_mlp(C.byref(dims), C.byref(io), C.byref(net),
                                                    capi.current_stream_ptr())
    if rc == -2:      # KVAE_E_UNSUPPORTED
        return None
    capi.check(rc)
    return m, P, mn, Pn, As, Cs, a_w, alpha


def _pack_mlp2_tensors(W1, b1, W2, b2):
    units = torch.cat([W1, b1.unsqueeze(1), W2.t()], dim=1)
    return torch.cat([units.reshape(-1), b2]).contiguous()


class MaskedFilterMLPFunction(torch.autograd.Function):
    """The masked filter with the two-layer perceptron weight network inside the recursion
    (kvae/state_space_model.py:219-352 with an observation mask) as ONE autograd node and one
    launch per direction:
    (a, mask, W1, b1, W2, b2, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R) ->
    (mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs, a_w, alpha (T+1,B,K)).
    The backward kernel returns the gradient w.r.t. the network's pre-softmax outputs; the
    gradients of W1, b1, W2, b2 are two small matrix products on the saved a_w."""

    @staticmethod
    def forward(ctx, a, mask, W1, b1, W2, b2, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R,
                symmetrize, burn_in, learn, first_weights):
        capi.require_gpu(a, mask, W1, b1, W2, b2, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R)
        a, mask, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R = map(
            _c, (a, mask, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R))
        T, B, p = a.shape
        K, n = A_K.shape[0], A_K.shape[1]
        hidden = W1.shape[0]
        packed = _pack_mlp2_tensors(W1.detach(), b1.detach(), W2.detach(), b2.detach())
        dims = _dims(T, B, n, p, K, a.dtype, symmetrize, burn_in)
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
        As, Cs, a_w = new(T + 1, B, n, n), new(T + 1, B, p, n), new(T, B, p)
        alpha = new(T + 1, B, K)
        alpha[0] = 0.0
        alpha[0, :, :first_weights] = 1.0 / first_weights
        io = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                       m_in=m_in, P_in=P_in, carry_per_sequence=int(m_in.dim() == 2),
                       mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs,
                       status=status_word(a.device))
        net = capi.fill(capi.MaskedMlp(), hidden=int(hidden), packed=packed, eps_z=eps_z, L_Q=L_Q,
                        eps_a=eps_a, L_R=L_R, a_w=a_w)
        capi.check(capi.load().kvae_filter_forward_masked_mlp(
            C.byref(dims), C.byref(io), C.byref(net), capi.current_stream_ptr()))
        ctx.save_for_backward(a, mask, W1, b1, W2, b2, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q,
                              eps_a, L_R, packed, m, P, mn, Pn, As, Cs, a_w, alpha)
        ctx.dims, ctx.learn = dims, bool(learn)
        ctx.set_materialize_grads(False)
        return m, P, mn, Pn, As, Cs, a_w, alpha

    @staticmethod
    def backward(ctx, g_m, g_P, g_mn, g_Pn, g_As, g_Cs, g_aw, g_alpha):
        (a, mask, W1, b1, W2, b2, A_K, C_K, Q, R, m_in, P_in, eps_z, L_Q, eps_a, L_R, packed,
         m, P, mn, Pn, As, Cs, a_w, alpha) = ctx.saved_tensors
        g_m, g_P, g_mn, g_Pn, g_As, g_Cs, g_aw, g_alpha = map(
            _c, (g_m, g_P, g_mn, g_Pn, g_As, g_Cs, g_aw, g_alpha))   # alive past the launch
        dims = ctx.dims
        T, B, n, p, K = dims.T, dims.B, dims.n, dims.p, dims.K
        new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
        d_a, d_alpha = new(T, B, p), new(T + 1, B, K)
        d_m_in, d_P_in = new(B, n), new(B, n, n)
        d_A_K, d_C_K, d_Q, d_R = new(K, n, n), new(K, p, n), new(n, n), new(p, p)
        d_logits = new(T, B, K)
        d_L_Q = new(n, n) if L_Q is not None else None
        d_L_R = new(p, p) if L_R is not None else None
        io = capi.FilterBwd()
        capi.fill(io.fwd, a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R, m_in=m_in,
                  P_in=P_in, carry_per_sequence=int(m_in.dim() == 2), mean_filt=m, cov_filt=P,
                  mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs)
        capi.fill(io, g_mean_filt=g_m, g_cov_filt=g_P, g_mean_pred=g_mn, g_cov_pred=g_Pn,
                  g_mat_As=g_As, g_mat_Cs=g_Cs, d_a=d_a, d_alpha=d_alpha, d_m_in=d_m_in,
                  d_P_in=d_P_in, d_A_K=d_A_K, d_C_K=d_C_K, d_Q=d_Q, d_R=d_R)
        nb = capi.MaskedMlpBwd()
        capi.fill(nb.fwd, hidden=int(W1.shape[0]), packed=packed, eps_z=eps_z, L_Q=L_Q,
                  eps_a=eps_a, L_R=L_R, a_w=a_w)
        capi.fill(nb, g_a_w=g_aw, g_alpha=g_alpha, d_logits=d_logits, d_L_Q=d_L_Q, d_L_R=d_L_R,
                  learn_weight_model=int(ctx.learn))
        lib = capi.load()
        nbytes = int(lib.kvae_filter_backward_masked_mlp_workspace_bytes(C.byref(dims)))
        ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=a.device)
        capi.check(lib.kvae_filter_backward_masked_mlp(C.byref(dims), C.byref(io), C.byref(nb),
                                                       ws.data_ptr(), nbytes,
                                                       capi.current_stream_ptr()))
        dW1 = db1 = dW2 = db2 = None
        if ctx.learn and any(ctx.needs_input_grad[2:6]):
            # parameters of the network: logits = W2 relu(W1 a_w + b1) + b2 over all (t, b)
            x = a_w.reshape(T * B, p)
            dl = d_logits.reshape(T * B, K)
            pre = torch.addmm(b1, x, W1.t())
            h = torch.relu(pre)
            dW2, db2 = dl.t() @ h, dl.sum(0)
            dpre = (dl @ W2) * (pre > 0).to(a.dtype)
            dW1, db1 = dpre.t() @ x, dpre.sum(0)
        if m_in.dim() == 1:
            d_m_in = d_m_in.sum(0) if ctx.needs_input_grad[10] else None
            d_P_in = d_P_in.sum(0) if ctx.needs_input_grad[11] else None
        return (d_a, None, dW1, db1, dW2, db2, d_A_K, d_C_K, d_Q, d_R, d_m_in, d_P_in, None, d_L_Q,
                None, d_L_R, None, None, None, None)


def masked_mlp_grad_supported(hidden, n, p, K, dtype):
    """Whether masked_filter_mlp_grad has a kernel for these dimensions."""
    dims2 = embedding_dims(n, p, K, dtype, fused=True)     # the lane-per-sequence shapes
    return dims2 is not None and masked_mlp_fits(hidden, dims2[1], dims2[2], dtype)


def masked_filter_mlp_grad(a, mask, first, last, A_K, C_K, Q, R, m_in, P_in, eps_z=None, L_Q=None,
                           eps_a=None, L_R=None, symmetrize=True, burn_in=0,
                           learn_weight_model=True):
    """The masked filter with the perceptron ``last(relu(first(.)))`` (two torch.nn.Linear)
    in the recursion, differentiable: see MaskedFilterMLPFunction.  Returns (mean_filt,
    cov_filt, mean_pred, cov_pred, mat_As, mat_Cs, a_w, alpha (T+1,B,K)), or None when no
    kernel covers the dimensions."""
    T, B, p = a.shape
    K, n = A_K.shape[0], A_K.shape[1]
    hidden = first.out_features
    dims2 = embedding_dims(n, p, K, a.dtype, fused=True)     # the lane-per-sequence shapes
    if dims2 is None or not masked_mlp_fits(hidden, dims2[1], dims2[2], a.dtype):
        return None
    W1, b1, W2, b2 = first.weight, first.bias, last.weight, last.bias
    if a.dtype != W1.dtype:
        W1, b1, W2, b2 = (t.to(a.dtype) for t in (W1, b1, W2, b2))
    flags = (bool(symmetrize), int(burn_in), bool(learn_weight_model), K)
    if dims2 == (n, p, K):
        return MaskedFilterMLPFunction.apply(a, mask, W1, b1, W2, b2, A_K, C_K, Q, R, m_in, P_in,
                                             eps_z, L_Q, eps_a, L_R, *flags)
    e = _Embedding(n, p, K, *dims2)
    pad = torch.nn.functional.pad
    W1p = pad(W1, (0, e.dp))                      # zero columns for the extra inputs
    W2p = pad(W2, (0, 0, 0, e.dK))                # extra outputs ...
    b2p = torch.cat([b2, b2.new_full((e.dK,), -1e30)])   # ... whose softmax weight is zero
    m, P, mn, Pn, As, Cs, a_w, alpha = MaskedFilterMLPFunction.apply(
        e.vec_p(a), mask, W1p, b1, W2p, b2p, e.bases_A(A_K), e.bases_C(C_K), e.cov_n(Q),
        e.cov_p(R), e.vec_n(m_in), e.cov_n(P_in), e.vec_n(eps_z), e.cov_n(L_Q), e.vec_p(eps_a),
        e.cov_p(L_R), *flags)
    return (e.cut_n(m), e.cut_nn(P), e.cut_n(mn), e.cut_nn(Pn), e.cut_nn(As), e.cut_pn(Cs),
            e.cut_p(a_w), e.cut_K(alpha))


FILTER_SMOOTH_OUTPUTS = ("mean_filt", "cov_filt", "mean_pred", "cov_pred", "mat_As", "mat_Cs",
                         "a_w", "alpha", "mean_smooth", "cov_smooth", "zs", "as_resampled")


@torch.no_grad()
def filter_smooth_fused(a, mask, A_K, C_K, Q, R, m_in, P_in, alpha=None, packed=None, hidden=0,
                        eps_z=None, L_Q=None, eps_a=None, L_R=None, eps_state=None, eps_obs=None,
                        symmetrize=True, want=("a_w", "as_resampled"), _first_weights=None):
    """Filter and RTS smoother in ONE launch that writes only the tensors named in ``want``
    (kvae_filter_smooth_fused; kvae/kalman_vae.py:141-160 as an inference consumer such as
    kvae/evaluation.py:100-140 uses it).  Weights: ``alpha`` (T+1,B,K) given, or the packed
    two-layer perceptron (``packed``, ``hidden``) evaluated inside the masked filter from
    uniform first weights.  Returns a dict of the requested tensors (names as in
    FILTER_SMOOTH_OUTPUTS; "a_w" only with the perceptron), or None when no kernel covers
    the dimensions."""
    capi.require_gpu(a, mask, A_K, C_K, Q, R, m_in, P_in, alpha, packed, eps_z, L_Q, eps_a, L_R,
                     eps_state, eps_obs)
    unknown = set(want) - set(FILTER_SMOOTH_OUTPUTS)
    if unknown:
        raise ValueError(f"unknown outputs {sorted(unknown)}")
    if (alpha is None) == (packed is None):
        raise ValueError("give either the mixing weights or the packed perceptron")
    T, B, p = a.shape
    K, n = A_K.shape[0], A_K.shape[1]
    dims2 = embedding_dims(n, p, K, a.dtype, fused=True)     # the lane-per-sequence shapes
    if dims2 is None:
        return None
    if dims2 != (n, p, K):
        e = _Embedding(n, p, K, *dims2)
        if packed is not None:
            if not masked_mlp_fits(hidden, e.p2, e.K2, a.dtype):
                return None
            packed = _pad_packed_mlp(e, packed, hidden)
        out = filter_smooth_fused(
            e.vec_p(a), mask, e.bases_A(A_K), e.bases_C(C_K), e.cov_n(Q), e.cov_p(R),
            e.vec_n(m_in), e.cov_n(P_in), e.weights(alpha), packed, hidden, e.vec_n(eps_z),
            e.cov_n(L_Q), e.vec_p(eps_a), e.cov_p(L_R), e.vec_n(eps_state), e.vec_p(eps_obs),
            symmetrize, want, _first_weights=K)
        if out is None:
            return None
        cut = {"mean_filt": e.cut_n, "cov_filt": e.cut_nn, "mean_pred": e.cut_n,
               "cov_pred": e.cut_nn, "mat_As": e.cut_nn, "mat_Cs": e.cut_pn, "a_w": e.cut_p,
               "alpha": e.cut_K, "mean_smooth": e.cut_n, "cov_smooth": e.cut_nn, "zs": e.cut_n,
               "as_resampled": e.cut_p}
        return {k: cut[k](v) for k, v in out.items()}
    return _filter_smooth_fused_native(a, mask, A_K, C_K, Q, R, m_in, P_in, alpha, packed, hidden,
                                       eps_z, L_Q, eps_a, L_R, eps_state, eps_obs, symmetrize,
                                       want, _first_weights or K)


def _filter_smooth_fused_native(a, mask, A_K, C_K, Q, R, m_in, P_in, alpha, packed, hidden, eps_z,
                                L_Q, eps_a, L_R, eps_state, eps_obs, symmetrize, want,
                                first_weights):
    (a, mask, A_K, C_K, Q, R, m_in, P_in, alpha, packed, eps_z, L_Q, eps_a, L_R, eps_state,
     eps_obs) = map(_c, (a, mask, A_K, C_K, Q, R, m_in, P_in, alpha, packed, eps_z, L_Q, eps_a,
                         L_R, eps_state, eps_obs))
    T, B, p = a.shape
    K, n = A_K.shape[0], A_K.shape[1]
    lib = capi.load()
    dims = _dims(T, B, n, p, K, a.dtype, symmetrize, 0)
    new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
    shapes = {"mean_filt": (T, B, n), "cov_filt": (T, B, n, n), "mean_pred": (T, B, n),
              "cov_pred": (T, B, n, n), "mat_As": (T + 1, B, n, n), "mat_Cs": (T + 1, B, p, n),
              "mean_smooth": (T, B, n), "cov_smooth": (T, B, n, n), "zs": (T, B, n),
              "as_resampled": (T, B, p)}
    out = {k: new(*shapes[k]) for k in want if k in shapes}
    net = None
    if packed is not None:
        # the perceptron steps from uniform weights over the real bases
        alpha = new(T + 1, B, K)
        alpha[0] = 0.0
        alpha[0, :, :first_weights] = 1.0 / first_weights
        a_w = new(T, B, p)
        net = capi.fill(capi.MaskedMlp(), hidden=int(hidden), packed=packed, eps_z=eps_z, L_Q=L_Q,
                        eps_a=eps_a, L_R=L_R, a_w=a_w)
        if "a_w" in want:
            out["a_w"] = a_w
    if "alpha" in want:
        out["alpha"] = alpha
    LR = L_R if eps_obs is not None else None
    st = status_word(a.device)
    f = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                  m_in=m_in, P_in=P_in, carry_per_sequence=int(m_in.dim() == 2),
                  mean_filt=out.get("mean_filt"), cov_filt=out.get("cov_filt"),
                  mean_pred=out.get("mean_pred"), cov_pred=out.get("cov_pred"),
                  mat_As=out.get("mat_As"), mat_Cs=out.get("mat_Cs"), status=st)
    s = capi.fill(capi.SmoothFwd(), LR=LR, eps_state=eps_state, eps_obs=eps_obs,
                  mean_smooth=out.get("mean_smooth"), cov_smooth=out.get("cov_smooth"),
                  zs=out.get("zs"), as_resampled=out.get("as_resampled"), status=st)
    nbytes = int(lib.kvae_filter_smooth_fused_workspace_bytes(C.byref(dims), C.byref(f)))
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=a.device)
    rc = lib.kvae_filter_smooth_fused(C.byref(dims), C.byref(f), C.byref(net) if net else None,
                                      C.byref(s), ws.data_ptr(), nbytes,
                                      capi.current_stream_ptr())
    if rc == -2:      # KVAE_E_UNSUPPORTED
        return None
    capi.check(rc)
    return out


def rollout_supported(n, p, K, dtype):
    """Whether kvae_predict_future has a kernel for these dimensions, directly or as a
    block of a larger compiled problem (small state dimensions only)."""
    dims = embedding_dims(n, p, K, dtype, fused=True)   # the lane-per-sequence shapes
    return dims is not None


@torch.no_grad()
def predict_rollout(phase, A_K, C_K, Q, mean_pred, cov_pred, mat_C=None, eps=None, packed=None,
                    hidden=0, alpha=None, sample=False, steps=1):
    """kvae_predict_future (kvae/state_space_model.py:486-575), values only.
    phase "all": every new step in one launch with the packed two-layer perceptron
    evaluated in the loop; returns (as_new (S,B,p), means, covs, mean_pred_new, cov_pred_new,
    mat_As_new, mat_Cs_new), each with S rows.  phase "emit": (as_new, means, covs) of one
    step; phase "advance": (mean_pred_new, cov_pred_new, mat_As_new, mat_Cs_new) of one step
    from the weights ``alpha`` (B,K)."""
    capi.require_gpu(A_K, C_K, Q, mean_pred, cov_pred, mat_C, eps, packed, alpha)
    K, n = A_K.shape[0], A_K.shape[1]
    p = C_K.shape[1]
    B = mean_pred.shape[0]
    dims2 = embedding_dims(n, p, K, mean_pred.dtype, fused=True)
    if dims2 is None:
        raise NotImplementedError(f"no rollout kernel covers z_dim={n}, a_dim={p}, K={K}")
    if dims2 != (n, p, K):
        e = _Embedding(n, p, K, *dims2)
        if packed is not None:
            packed = _pad_packed_mlp(e, packed, hidden)
        out = predict_rollout(phase, e.bases_A(A_K), e.bases_C(C_K), e.cov_n(Q),
                              e.vec_n(mean_pred), e.cov_n(cov_pred), e.mat_pn(mat_C),
                              e.vec_n(eps), packed, hidden, e.weights(alpha), sample, steps)
        cut = {"all": (e.cut_p, e.cut_n, e.cut_nn, e.cut_n, e.cut_nn, e.cut_nn, e.cut_pn),
               "emit": (e.cut_p, e.cut_n, e.cut_nn),
               "advance": (e.cut_n, e.cut_nn, e.cut_nn, e.cut_pn)}[phase]
        return tuple(f(t) for f, t in zip(cut, out))
    S = int(steps) if phase == "all" else 1
    dt, dev = mean_pred.dtype, mean_pred.device
    A_K, C_K, Q, mean_pred, cov_pred, mat_C, eps, packed, alpha = map(
        _c, (A_K, C_K, Q, mean_pred, cov_pred, mat_C, eps, packed, alpha))
    new = lambda *s: torch.empty(s, dtype=dt, device=dev)
    emits, advances = phase != "advance", phase != "emit"
    as_new = new(S, B, p) if emits else None
    means, covs = (new(S, B, n), new(S, B, n, n)) if emits else (None, None)
    mn2, Pn2 = (new(S, B, n), new(S, B, n, n)) if advances else (None, None)
    As2, Cs2 = (new(S, B, n, n), new(S, B, p, n)) if advances else (None, None)
    dims = capi.Dims(S, B, n, p, K, capi.dtype_code(dt), 1, 0)
    io = capi.fill(capi.Rollout(), A_K=A_K, C_K=C_K, Q=Q, mean_pred=mean_pred, cov_pred=cov_pred,
                   mat_C=mat_C, eps=eps if sample else None, packed_mlp=packed, alpha=alpha,
                   hidden=int(hidden),
                   phase={"all": capi.ROLLOUT_ALL, "emit": capi.ROLLOUT_EMIT,
                          "advance": capi.ROLLOUT_ADVANCE}[phase],
                   sample=int(bool(sample)), as_new=as_new, means=means, covs=covs,
                   mean_pred_new=mn2, cov_pred_new=Pn2, mat_As_new=As2, mat_Cs_new=Cs2,
                   status=status_word(dev))
    capi.check(capi.load().kvae_predict_future(C.byref(dims), C.byref(io),
                                               capi.current_stream_ptr()))
    if phase == "all":
        return as_new, means, covs, mn2, Pn2, As2, Cs2
    if phase == "emit":
        return as_new, means, covs
    return mn2, Pn2, As2, Cs2


def kalman_filter_steps(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, symmetrize=True, burn_in=0,
                        want_mats=True):
    e = _embedding(A_K.shape[1], a.shape[-1], A_K.shape[0], a.dtype)
    if e.identity:
        return KalmanFilterFunction.apply(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, symmetrize,
                                          burn_in, want_mats)
    m, P, mn, Pn, As, Cs = KalmanFilterFunction.apply(
        e.vec_p(a), e.weights(alpha), mask, e.bases_A(A_K), e.bases_C(C_K), e.cov_n(Q),
        e.cov_p(R), e.vec_n(m_in), e.cov_n(P_in), symmetrize, burn_in, want_mats)
    if want_mats:
        As, Cs = e.cut_nn(As), e.cut_pn(Cs)
    return e.cut_n(m), e.cut_nn(P), e.cut_n(mn), e.cut_nn(Pn), As, Cs


def kalman_smooth_steps(m, P, mn, Pn, As, Cs, LR=None, eps_state=None, eps_obs=None,
                        symmetrize=True, burn_in=0):
    e = _embedding(m.shape[-1], Cs.shape[-2], None, m.dtype)
    if e.identity:
        return KalmanSmoothFunction.apply(m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs,
                                          symmetrize, burn_in)
    ms, Ps, zs, ares = KalmanSmoothFunction.apply(
        e.vec_n(m), e.cov_n(P), e.vec_n(mn), e.cov_n(Pn), e.mat_nn(As), e.mat_pn(Cs), e.cov_p(LR),
        e.vec_n(eps_state), e.vec_p(eps_obs), symmetrize, burn_in)
    return e.cut_n(ms), e.cut_nn(Ps), e.cut_n(zs), e.cut_p(ares)


def _elbo_operands(e, a, ms, Ps, As, Cs, Q, R, m0, P0, eps):
    return (e.vec_p(a), e.vec_n(ms), e.cov_n(Ps), e.mat_nn(As), e.mat_pn(Cs), e.cov_n(Q),
            e.cov_p(R), e.vec_n(m0), e.cov_n(P0), e.vec_n(eps))


def kalman_elbo_terms(a, ms, Ps, As, Cs, Q, R, m0, P0, eps):
    e = _embedding(ms.shape[-1], a.shape[-1], None, a.dtype)
    if e.identity:
        return KalmanElboFunction.apply(a, ms, Ps, As, Cs, Q, R, m0, P0, eps)
    obs, trans, post, z = KalmanElboFunction.apply(
        *_elbo_operands(e, a, ms, Ps, As, Cs, Q, R, m0, P0, eps))
    return obs + e.c_obs, trans + e.c_lat, post + e.c_lat, e.cut_n(z)


def kalman_elbo_objective(a, ms, Ps, As, Cs, Q, R, m0, P0, eps, w_obs, w_trans, w_post):
    """(objective, weighted totals (3,), terms (3,T,B), z): see KalmanElboObjectiveFunction."""
    w = (float(w_obs), float(w_trans), float(w_post))
    e = _embedding(ms.shape[-1], a.shape[-1], None, a.dtype)
    if e.identity:
        return KalmanElboObjectiveFunction.apply(a, ms, Ps, As, Cs, Q, R, m0, P0, eps, *w)
    objective, weighted, terms, z = KalmanElboObjectiveFunction.apply(
        *_elbo_operands(e, a, ms, Ps, As, Cs, Q, R, m0, P0, eps), *w)
    objective, weighted, terms = _shift_totals(
        objective, weighted, terms, a.shape[0] * a.shape[1], ((e.c_obs, e.c_lat, e.c_lat), w), a)
    return objective, weighted, terms, e.cut_n(z)


@torch.no_grad()
def kalman_filter_smooth_nograd(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, LR=None,
                                eps_state=None, eps_obs=None, symmetrize=True):
    """Filter then smoother through the single C-ABI call kvae_filter_smooth_forward
    (inference / imputation: no autograd graph is recorded).  Returns
    (mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs,
     mean_smooth, cov_smooth, zs, as_resampled)."""
    e = _embedding(A_K.shape[1], a.shape[-1], A_K.shape[0], a.dtype)
    if e.identity:
        return _filter_smooth_native(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, LR, eps_state,
                                     eps_obs, symmetrize)
    m, P, mn, Pn, As, Cs, ms, Ps, zs, ares = _filter_smooth_native(
        e.vec_p(a), e.weights(alpha), mask, e.bases_A(A_K), e.bases_C(C_K), e.cov_n(Q),
        e.cov_p(R), e.vec_n(m_in), e.cov_n(P_in), e.cov_p(LR), e.vec_n(eps_state),
        e.vec_p(eps_obs), symmetrize)
    return (e.cut_n(m), e.cut_nn(P), e.cut_n(mn), e.cut_nn(Pn), e.cut_nn(As), e.cut_pn(Cs),
            e.cut_n(ms), e.cut_nn(Ps), e.cut_n(zs), e.cut_p(ares))


def _filter_smooth_native(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, LR, eps_state, eps_obs,
                          symmetrize):
    capi.require_gpu(a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, LR, eps_state, eps_obs)
    a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, LR, eps_state, eps_obs = map(
        _c, (a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, LR, eps_state, eps_obs))
    T, B, p = a.shape
    K, n = A_K.shape[0], A_K.shape[1]
    dims = _dims(T, B, n, p, K, a.dtype, symmetrize, 0)
    new = lambda *s: torch.empty(s, dtype=a.dtype, device=a.device)
    m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
    As, Cs = new(T + 1, B, n, n), new(T + 1, B, p, n)
    ms, Ps, zs, ares = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, p)
    st = status_word(a.device)
    f = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                  m_in=m_in, P_in=P_in, carry_per_sequence=int(m_in.dim() == 2), mean_filt=m,
                  cov_filt=P, mean_pred=mn, cov_pred=Pn, mat_As=As, mat_Cs=Cs, status=st)
    s = capi.fill(capi.SmoothFwd(), mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn,
                  mat_As=As, mat_Cs=Cs, LR=LR, eps_state=eps_state, eps_obs=eps_obs,
                  mean_smooth=ms, cov_smooth=Ps, zs=zs, as_resampled=ares, status=st)
    capi.check(capi.load().kvae_filter_smooth_forward(C.byref(dims), C.byref(f), C.byref(s),
                                                      capi.current_stream_ptr()))
    return m, P, mn, Pn, As, Cs, ms, Ps, zs, ares
