"""Drive the six kernel entry points with torch tensors, through either
  * backend "hip":     the product library libkvae_kalman.so (CUDA tensors), or
  * backend "hostsim": tests/hostsim/libkvae_hostsim.so, the test-only host
                       build of the same per-sequence source (CPU tensors).
Both take the structs of include/kvae_kalman.h, so one harness serves both and
the GPU parity tests and the CPU math tests share every comparison."""
import ctypes as C
import os
import subprocess

import torch

from kalman_vae_amd import _capi as capi

HERE = os.path.dirname(os.path.abspath(__file__))
HOSTSIM_DIR = os.path.join(HERE, "hostsim")
HOSTSIM_LIB = os.path.join(HOSTSIM_DIR, "libkvae_hostsim.so")

_hostsim = None


def hostsim_lib():
    global _hostsim
    if _hostsim is None:
        src = os.path.join(HOSTSIM_DIR, "hostsim.cpp")
        csrc = os.path.join(HERE, "..", "kalman_vae_amd", "csrc")
        deps = [src, os.path.join(csrc, "kalman_core.h"), os.path.join(csrc, "kalman_wave.h"),
                os.path.join(csrc, "kalman_loss.h"), os.path.join(csrc, "kalman_row.h")]
        stale = (not os.path.exists(HOSTSIM_LIB)
                 or os.path.getmtime(HOSTSIM_LIB) < max(os.path.getmtime(f) for f in deps))
        if stale:
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC",
                                   "-o", HOSTSIM_LIB, src])
        _hostsim = C.CDLL(HOSTSIM_LIB)
        capi.declare(_hostsim, prefix="hostsim_", with_stream=False)
        capi.declare(_hostsim, prefix="hostsim_wave_", with_stream=False)
        for name, io in (("forward", capi.LossFwd), ("backward", capi.LossBwd)):
            fn = getattr(_hostsim, "hostsim_row_kalman_loss_" + name)
            fn.argtypes = [C.POINTER(capi.Dims), C.POINTER(io)]
            fn.restype = C.c_int
    return _hostsim


class Backend:
    def __init__(self, kind):
        """kind: "hip" (product library; picks the lane- or wavefront-per-sequence
        kernels by dimension), "hostsim" (host build of the lane source) or
        "hostsim_wave" (host build of the wavefront-per-sequence source), "hostsim_row"
        (host build of the 16-lanes-per-sequence source: fused objective only)."""
        assert kind in ("hip", "hostsim", "hostsim_wave", "hostsim_row")
        self.kind = kind
        self.device = torch.device("cuda:0" if kind == "hip" else "cpu")
        self.lib = capi.load() if kind == "hip" else hostsim_lib()

    def _call(self, name, dims, io, op=None):
        if self.kind != "hip":
            code = getattr(self.lib, self.kind + "_" + name)(C.byref(dims), C.byref(io))
            assert code == 0, code
            return
        stream = torch.cuda.current_stream().cuda_stream
        fn = getattr(self.lib, "kvae_" + name)
        if op is None:
            code = fn(C.byref(dims), C.byref(io), stream)
        else:
            nbytes = self.lib.kvae_workspace_bytes(C.byref(dims), op)
            ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=self.device)
            code = fn(C.byref(dims), C.byref(io), ws.data_ptr(), nbytes, stream)
            torch.cuda.synchronize()
        assert code == 0, capi.load().kvae_error_string(code)

    def dims(self, T, B, n, p, K, dtype, symmetrize=True, burn_in=0):
        return capi.Dims(T, B, n, p, K, capi.dtype_code(dtype), int(symmetrize), burn_in)

    def new(self, *shape, dtype):
        # NaN-filled so that an element a kernel forgets to write is noticed
        return torch.full(shape, float("nan"), dtype=dtype, device=self.device)

    def status(self):
        return torch.zeros(1, dtype=torch.int32, device=self.device)

    # -- filter -------------------------------------------------------------
    def filter_forward(self, a, alpha, mask, A_K, C_K, Q, R, m_in, P_in, symmetrize=True,
                       want_mats=True):
        T, B, p = a.shape
        n, K = A_K.shape[1], A_K.shape[0]
        dt = a.dtype
        d = self.dims(T, B, n, p, K, dt, symmetrize)
        out = dict(mean_filt=self.new(T, B, n, dtype=dt), cov_filt=self.new(T, B, n, n, dtype=dt),
                   mean_pred=self.new(T, B, n, dtype=dt), cov_pred=self.new(T, B, n, n, dtype=dt))
        if want_mats:
            out["mat_As"] = self.new(T + 1, B, n, n, dtype=dt)
            out["mat_Cs"] = self.new(T + 1, B, p, n, dtype=dt)
        st = self.status()
        io = capi.fill(capi.FilterFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                       m_in=m_in, P_in=P_in, carry_per_sequence=int(m_in.dim() == 2),
                       status=st, **out)
        self._call("filter_forward", d, io)
        out["status"] = st
        out["_io"] = io
        out["_keep"] = (a, alpha, mask, A_K, C_K, Q, R, m_in, P_in)
        return out

    def filter_backward(self, fwd, grads, symmetrize=True, burn_in=0):
        a, alpha, mask, A_K, C_K, Q, R, m_in, P_in = fwd["_keep"]
        T, B, p = a.shape
        n, K = A_K.shape[1], A_K.shape[0]
        dt = a.dtype
        d = self.dims(T, B, n, p, K, dt, symmetrize, burn_in)
        out = dict(d_a=self.new(T, B, p, dtype=dt), d_alpha=self.new(T + 1, B, K, dtype=dt),
                   d_m_in=self.new(B, n, dtype=dt), d_P_in=self.new(B, n, n, dtype=dt),
                   d_A_K=self.new(K, n, n, dtype=dt), d_C_K=self.new(K, p, n, dtype=dt),
                   d_Q=self.new(n, n, dtype=dt), d_R=self.new(p, p, dtype=dt))
        io = capi.FilterBwd()
        io.fwd = fwd["_io"]
        capi.fill(io, **{"g_" + k: v for k, v in grads.items()}, **out)
        self._call("filter_backward", d, io, capi.OP_FILTER_BWD)
        return out

    # -- smoother -----------------------------------------------------------
    def smooth_forward(self, m, P, mn, Pn, As, Cs, LR=None, eps_state=None, eps_obs=None,
                       symmetrize=True):
        T, B, n = m.shape
        p = Cs.shape[2]
        dt = m.dtype
        d = self.dims(T, B, n, p, 0, dt, symmetrize)
        d.K = self._any_K(n, p)
        out = dict(mean_smooth=self.new(T, B, n, dtype=dt), cov_smooth=self.new(T, B, n, n, dtype=dt),
                   zs=self.new(T, B, n, dtype=dt), as_resampled=self.new(T, B, p, dtype=dt))
        st = self.status()
        io = capi.fill(capi.SmoothFwd(), mean_filt=m, cov_filt=P, mean_pred=mn, cov_pred=Pn,
                       mat_As=As, mat_Cs=Cs, LR=LR, eps_state=eps_state, eps_obs=eps_obs,
                       status=st, **out)
        self._call("smooth_forward", d, io)
        out["status"] = st
        out["_io"] = io
        out["_keep"] = (m, P, mn, Pn, As, Cs, LR, eps_state, eps_obs)
        return out

    def smooth_backward(self, fwd, grads, symmetrize=True, burn_in=0, want_LR=True):
        m, P, mn, Pn, As, Cs = fwd["_keep"][:6]
        T, B, n = m.shape
        p = Cs.shape[2]
        dt = m.dtype
        d = self.dims(T, B, n, p, self._any_K(n, p), dt, symmetrize, burn_in)
        out = dict(d_mean_filt=self.new(T, B, n, dtype=dt), d_cov_filt=self.new(T, B, n, n, dtype=dt),
                   d_mean_pred=self.new(T, B, n, dtype=dt), d_cov_pred=self.new(T, B, n, n, dtype=dt),
                   d_mat_As=self.new(T + 1, B, n, n, dtype=dt),
                   d_mat_Cs=self.new(T + 1, B, p, n, dtype=dt))
        if want_LR:
            out["d_LR"] = self.new(p, p, dtype=dt)
        io = capi.SmoothBwd()
        io.fwd = fwd["_io"]
        capi.fill(io, **{"g_" + k: v for k, v in grads.items()}, **out)
        self._call("smooth_backward", d, io, capi.OP_SMOOTH_BWD)
        return out

    # -- ELBO ---------------------------------------------------------------
    def elbo_forward(self, a, ms, Ps, As, Cs, Q, R, m0, P0, eps):
        T, B, n = ms.shape
        p = a.shape[2]
        dt = a.dtype
        d = self.dims(T, B, n, p, self._any_K(n, p), dt)
        out = dict(obs=self.new(T, B, dtype=dt), trans=self.new(T, B, dtype=dt),
                   post=self.new(T, B, dtype=dt), z=self.new(T, B, n, dtype=dt))
        st = self.status()
        io = capi.fill(capi.ElboFwd(), a=a, mean_smooth=ms, cov_smooth=Ps, mat_As=As, mat_Cs=Cs,
                       Q=Q, R=R, m0=m0, P0=P0, eps=eps, status=st, **out)
        self._call("elbo_forward", d, io)
        out["status"] = st
        out["_io"] = io
        out["_keep"] = (a, ms, Ps, As, Cs, Q, R, m0, P0, eps)
        return out

    def elbo_backward(self, fwd, g_obs, g_trans, g_post):
        a, ms, Ps, As, Cs = fwd["_keep"][:5]
        T, B, n = ms.shape
        p = a.shape[2]
        dt = a.dtype
        d = self.dims(T, B, n, p, self._any_K(n, p), dt)
        out = dict(d_a=self.new(T, B, p, dtype=dt), d_mean_smooth=self.new(T, B, n, dtype=dt),
                   d_cov_smooth=self.new(T, B, n, n, dtype=dt),
                   d_mat_As=self.new(T + 1, B, n, n, dtype=dt),
                   d_mat_Cs=self.new(T + 1, B, p, n, dtype=dt),
                   d_Q=self.new(n, n, dtype=dt), d_R=self.new(p, p, dtype=dt),
                   d_m0=self.new(n, dtype=dt), d_P0=self.new(n, n, dtype=dt))
        io = capi.ElboBwd()
        io.fwd = fwd["_io"]
        capi.fill(io, g_obs=g_obs, g_trans=g_trans, g_post=g_post, **out)
        self._call("elbo_backward", d, io, capi.OP_ELBO_BWD)
        return out

    # -- fused objective: filter -> smoother -> ELBO terms in one call -----------
    def loss_forward(self, a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, weights,
                     symmetrize=True, want_terms=True):
        T, B, p = a.shape
        n, K = A_K.shape[1], A_K.shape[0]
        dt = a.dtype
        d = self.dims(T, B, n, p, K, dt, symmetrize)
        lib = capi.load() if self.kind == "hip" else None
        esize = a.element_size()
        saved_bytes = (lib.kvae_loss_saved_bytes(C.byref(d)) if lib
                       else T * B * (3 * n + 3 * n * n) * esize)
        saved = self.new(saved_bytes // esize, dtype=dt)
        out = dict(out=self.new(4, dtype=dt))
        if want_terms:
            out["terms"] = self.new(3, T, B, dtype=dt)
        st = self.status()
        io = capi.fill(capi.LossFwd(), a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R,
                       m0=m0, P0=P0, eps=eps, saved=saved, status=st, **out)
        io.w_obs, io.w_trans, io.w_post = weights
        self._call("kalman_loss_forward", d, io, capi.OP_LOSS_FWD)
        out.update(status=st, saved=saved, _io=io,
                   _keep=(a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps))
        return out

    def loss_backward(self, fwd, g_scale, stride, symmetrize=True, burn_in=0):
        a, alpha, mask, A_K, C_K = fwd["_keep"][:5]
        T, B, p = a.shape
        n, K = A_K.shape[1], A_K.shape[0]
        dt = a.dtype
        d = self.dims(T, B, n, p, K, dt, symmetrize, burn_in)
        out = dict(d_a=self.new(T, B, p, dtype=dt), d_alpha=self.new(T + 1, B, K, dtype=dt),
                   d_A_K=self.new(K, n, n, dtype=dt), d_C_K=self.new(K, p, n, dtype=dt),
                   d_Q=self.new(n, n, dtype=dt), d_R=self.new(p, p, dtype=dt),
                   d_m0=self.new(n, dtype=dt), d_P0=self.new(n, n, dtype=dt))
        io = capi.LossBwd()
        io.fwd = fwd["_io"]
        capi.fill(io, g_scale=g_scale, g_scale_stride=stride, **out)
        self._call("kalman_loss_backward", d, io, capi.OP_LOSS_BWD)
        out["_keep"] = g_scale
        return out

    def loss_value_and_grad(self, a, alpha, mask, A_K, C_K, Q, R, m0, P0, eps, weights, g_scale,
                            stride, symmetrize=True, burn_in=0, want_terms=True):
        """kvae_kalman_loss_value_and_grad: the outputs of both calls above from one call."""
        assert self.kind == "hip"
        T, B, p = a.shape
        n, K = A_K.shape[1], A_K.shape[0]
        dt = a.dtype
        d = self.dims(T, B, n, p, K, dt, symmetrize, burn_in)
        saved = self.new(self.lib.kvae_loss_saved_bytes(C.byref(d)) // a.element_size(), dtype=dt)
        out = dict(out=self.new(4, dtype=dt))
        if want_terms:
            out["terms"] = self.new(3, T, B, dtype=dt)
        grads = dict(d_a=self.new(T, B, p, dtype=dt), d_alpha=self.new(T + 1, B, K, dtype=dt),
                     d_A_K=self.new(K, n, n, dtype=dt), d_C_K=self.new(K, p, n, dtype=dt),
                     d_Q=self.new(n, n, dtype=dt), d_R=self.new(p, p, dtype=dt),
                     d_m0=self.new(n, dtype=dt), d_P0=self.new(n, n, dtype=dt))
        st = self.status()
        io = capi.LossBwd()
        capi.fill(io.fwd, a=a, alpha=alpha, mask=mask, A_K=A_K, C_K=C_K, Q=Q, R=R, m0=m0, P0=P0,
                  eps=eps, saved=saved, status=st, **out)
        io.fwd.w_obs, io.fwd.w_trans, io.fwd.w_post = weights
        capi.fill(io, g_scale=g_scale, g_scale_stride=stride, **grads)
        self._call("kalman_loss_value_and_grad", d, io, capi.OP_LOSS_VALUE_AND_GRAD)
        out.update(grads)
        out.update(status=st, _keep=(saved, g_scale))
        return out

    @staticmethod
    def _any_K(n, p):
        """The smoother and ELBO kernels do not depend on K, but the dispatch
        table is keyed on (n,p,K): pick the K compiled for this (n,p)."""
        return {(4, 2): 3, (2, 1): 2, (3, 2): 2, (8, 4): 4,
                (32, 16): 8, (16, 8): 4, (6, 3): 2}[(n, p)]


# -- random problem instances ------------------------------------------------
def random_problem(T, B, n, p, K, dtype, seed=0, mask_kind="none", general=True):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    prob = {}
    prob["a"] = 0.7 * rn(T, B, p)
    prob["alpha"] = torch.softmax(rn(T + 1, B, K), dim=-1)
    prob["A_K"] = 0.85 * torch.eye(n, dtype=torch.float64) + 0.12 * rn(K, n, n)
    prob["C_K"] = 0.85 * torch.eye(p, n, dtype=torch.float64) + 0.12 * rn(K, p, n)
    if general:
        WQ, WR, W0 = rn(n, n), rn(p, p), rn(n, n)
        prob["Q"] = 0.4 * torch.eye(n, dtype=torch.float64) + 0.08 * WQ @ WQ.T
        prob["R"] = 0.3 * torch.eye(p, dtype=torch.float64) + 0.08 * WR @ WR.T
        prob["m0"] = 0.3 * rn(n)
        prob["P0"] = 0.6 * torch.eye(n, dtype=torch.float64) + 0.1 * W0 @ W0.T
    else:
        prob["Q"] = torch.eye(n, dtype=torch.float64)
        prob["R"] = torch.eye(p, dtype=torch.float64)
        prob["m0"] = torch.zeros(n, dtype=torch.float64)
        prob["P0"] = torch.eye(n, dtype=torch.float64)
    if mask_kind == "none":
        prob["mask"] = None
    elif mask_kind == "binary":
        mk = (torch.rand(T, B, generator=g) > 0.3).double()
        mk[0] = 1.0
        mk[-1] = 1.0
        prob["mask"] = mk
    else:
        prob["mask"] = torch.rand(T, B, generator=g).double()
    prob["eps_s"] = rn(T, B, n)
    prob["eps_o"] = rn(T, B, p)
    prob["eps_e"] = rn(T, B, n)
    return {k: (v.to(dtype) if v is not None else None) for k, v in prob.items()}


class AlphaReplay:
    """weight_fn for the oracle that replays precomputed mixing weights."""

    def __init__(self, alpha):
        self.alpha = alpha
        self.t = 0

    def __call__(self, a_w):
        self.t += 1
        return self.alpha[self.t].unsqueeze(0)
