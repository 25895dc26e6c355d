"""Drop-in for the reference's ``StateSpaceModel`` whose filter and smoother run
as HIP kernels on MI355X.

Same constructor arguments, parameter names (``a_0, _mat_A_K, _mat_C_K,
_mat_Q_L, _mat_R_L, weight_model.*`` -- reference checkpoints load), matrix
properties/setters and the same ``kalman_filter`` / ``kalman_smooth``
signatures and return tuples as kvae/state_space_model.py:15-484.

How the work is split:
  * The dynamics-parameter network stays a PyTorch module.  Without an
    observation mask its inputs are the observations themselves, so all mixing
    weights are produced by ONE network call over the whole sequence and the
    T-step recursion runs inside a single kernel launch.
  * With a mask the network input at step t depends on the filter state
    (state_space_model.py:232-282), so the recursion is cut into one-step
    kernel segments with the imputation and the network call in between.
  * Random draws are made with the same primitive and in the same order as
    the reference's ``MultivariateNormal.rsample`` calls, so a seeded run
    consumes the generator identically (``rng_parity``; switch it off to draw
    one tensor per call instead of two per step).
"""
import contextlib
from typing import Optional

import torch
import torch.nn as nn
from torch.distributions.utils import _standard_normal

from . import functions as F
from . import masked_lstm
from .dynamics_parameter_network import make_weight_model
from .misc import _validate_shape
from .sample_control import SampleControl, wants_noise


class StateSpaceModel(nn.Module):
    def __init__(
        self,
        a_dim: int,
        z_dim: int,
        K: int,
        dynamics_parameter_network: str,
        hidden_dim: int = 128,
        num_layers: int = 2,
        Q_reg: float = 1e-3,
        R_reg: float = 1e-3,
        init_transition_reg_weight: float = 0.9,
        init_observation_reg_weight: float = 0.9,
        initial_state_mean: Optional[torch.Tensor] = None,
        initial_state_covariance: Optional[torch.Tensor] = None,
        fix_matrices: bool = False,
        learn_noise_covariance: bool = False,
        init_noise_scale: float = 1.0,
    ):
        super().__init__()
        self.a_dim, self.z_dim, self.K = a_dim, z_dim, K
        self.a_0 = nn.Parameter(torch.zeros(a_dim))
        wa, wc = init_transition_reg_weight, init_observation_reg_weight
        self._mat_A_K = nn.Parameter((1.0 - wa) * torch.randn(K, z_dim, z_dim)
                                     + wa * torch.eye(z_dim))
        self._mat_C_K = nn.Parameter((1.0 - wc) * torch.randn(K, a_dim, z_dim)
                                     + wc * torch.eye(a_dim, z_dim))
        # the *unconstrained* square factors are the parameters; Q = L L^T + reg I
        # (torch.linalg.cholesky returns a column-major tensor: stored row-major here, so
        # that the kernels read the parameters in place instead of through a copy per step)
        self._mat_Q_L = nn.Parameter(torch.linalg.cholesky(init_noise_scale * torch.eye(z_dim)).contiguous())
        self._mat_R_L = nn.Parameter(torch.linalg.cholesky(init_noise_scale * torch.eye(a_dim)).contiguous())
        self.learn_noise_covariance = learn_noise_covariance
        self.fix_matrices = fix_matrices
        self.Q_reg, self.R_reg = Q_reg, R_reg
        # plain tensors (not buffers: they must stay out of the state_dict)
        self._a_eye = torch.eye(a_dim)
        self._z_eye = torch.eye(z_dim)
        if initial_state_mean is None:
            initial_state_mean = torch.zeros(z_dim)
        else:
            _validate_shape(initial_state_mean, (z_dim,), "initial_state_mean")
        if initial_state_covariance is None:
            initial_state_covariance = torch.eye(z_dim)
        else:
            _validate_shape(initial_state_covariance, (z_dim, z_dim), "initial_state_covariance")
        self.initial_state_mean = initial_state_mean
        self.initial_state_covariance = initial_state_covariance
        self.weight_model = make_weight_model(dynamics_parameter_network, a_dim, K,
                                              hidden_dim, num_layers)
        # behaviour switches that have no counterpart in the reference
        self._matrix_cache = None   # see cached_matrices()
        self._noise_pool = None     # see _draw()
        self._draw_log = None       # see _recording_draws()
        self._defer_status = False  # kalman_elbo reads the status word once per step
        self.rng_parity = True      # draw noise exactly like the reference does
        # kalman_elbo: compute the objective with the one-launch-per-direction kernels that
        # never materialise the filter / smoother tensors (info then holds None for them)
        self.fused_loss = False
        self.fused_loss_terms = False   # also return the per-(t,b) terms from that path
        # masked inference with a two-layer perceptron network: evaluate it inside the
        # filter kernel (False: step the network between one-step filter launches)
        self.in_kernel_weight_network = True
        self.check_numerics = True  # raise on a non-positive-definite covariance (host sync)

    def _apply(self, fn):
        super()._apply(fn)
        for name in ("_a_eye", "_z_eye", "initial_state_mean", "initial_state_covariance"):
            setattr(self, name, fn(getattr(self, name)))
        return self

    # -- matrices (state_space_model.py:112-170) ------------------------------
    def _gated(self, matrix, noise):
        if self.fix_matrices or (noise and not self.learn_noise_covariance):
            matrix = matrix.detach()
        return matrix

    @property
    def mat_Q(self):
        return self._noise_covariance("Q", self._mat_Q_L, self._z_eye, self.Q_reg)

    @property
    def mat_R(self):
        return self._noise_covariance("R", self._mat_R_L, self._a_eye, self.R_reg)

    def _noise_covariance(self, key, factor, eye, reg):
        cache = self._matrix_cache
        if cache is not None and key in cache:
            return cache[key]
        value = self._gated(factor @ factor.T + eye * reg, True)
        if cache is not None:
            cache[key] = value
        return value

    @contextlib.contextmanager
    def cached_matrices(self):
        """Within the block ``mat_Q`` / ``mat_R`` (and the Cholesky factor of R) are
        formed once and reused: the reference rebuilds them on every property
        access (state_space_model.py:112-146), a handful of tiny kernels each time.
        Only valid while the parameters are not modified, i.e. inside one step."""
        outer = self._matrix_cache
        self._matrix_cache = {} if outer is None else outer
        noise_is_constant = (self.fix_matrices or not self.learn_noise_covariance
                             or not torch.is_grad_enabled())
        if (outer is None and noise_is_constant and self._mat_Q_L.is_cuda
                and max(self.z_dim, self.a_dim) <= 32):
            # no gradient flows into the noise factors: Q, R and chol(R) in one launch
            Q, R, L_R = F.noise_factors(self._mat_Q_L, self._mat_R_L, self.Q_reg, self.R_reg)
            self._matrix_cache.update(Q=Q, R=R, L_R=L_R)
        try:
            yield self
        finally:
            self._matrix_cache = outer

    def _cholesky_R(self):
        cache = self._matrix_cache
        if cache is not None and "L_R" in cache:
            return cache["L_R"]
        value = F.cholesky_factor(self.mat_R)
        if cache is not None:
            cache["L_R"] = value
        return value

    @property
    def mat_A_K(self):
        return self._gated(self._mat_A_K, False)

    @property
    def mat_C_K(self):
        return self._gated(self._mat_C_K, False)

    @mat_Q.setter
    def mat_Q(self, value):
        _validate_shape(value, (self.z_dim, self.z_dim), "mat_Q")
        self._mat_Q_L = nn.Parameter(torch.linalg.cholesky(value).contiguous())
        self.Q_reg = 0.0

    @mat_R.setter
    def mat_R(self, value):
        _validate_shape(value, (self.a_dim, self.a_dim), "mat_R")
        self._mat_R_L = nn.Parameter(torch.linalg.cholesky(value).contiguous())
        self.R_reg = 0.0

    @mat_A_K.setter
    def mat_A_K(self, value):
        _validate_shape(value, (self.K, self.z_dim, self.z_dim), "mat_A_K")
        self._mat_A_K = nn.Parameter(value)

    @mat_C_K.setter
    def mat_C_K(self, value):
        _validate_shape(value, (self.K, self.a_dim, self.z_dim), "mat_C_K")
        self._mat_C_K = nn.Parameter(value)

    # -- helpers --------------------------------------------------------------
    def _modes(self, sample_control, filtering):
        state = wants_noise(sample_control.state_transition, "state_transition")
        obs = wants_noise(sample_control.observation, "observation")
        if filtering and self.training:
            # "mean" is an evaluation-only mode of the filter (:237-240, :256-259)
            if not state:
                raise ValueError("sample_control.state_transition must be 'sample' for training")
            if not obs:
                raise ValueError("sample_control.observation must be 'sample' for training")
        return state, obs

    def _draw(self, shape, like):
        """A standard-normal tensor; served from the pre-drawn pool (one launch for all
        the noise of a step, see kalman_elbo) when one is installed."""
        pool = self._noise_pool
        if pool:
            assert tuple(pool[0].shape) == tuple(shape), (tuple(pool[0].shape), tuple(shape))
            eps = pool.pop(0)
        else:
            eps = _standard_normal(shape, dtype=like.dtype, device=like.device)
        if self._draw_log is not None:
            self._draw_log.append(eps)
        return eps

    @contextlib.contextmanager
    def _recording_draws(self):
        """Record what _draw hands out inside the block (a fused path draws all its noise
        before it learns whether a kernel covers the call); the record is the block's value.
        Recording stops when the block ends, however it ends."""
        log = self._draw_log = []
        try:
            yield log
        finally:
            self._draw_log = None

    def _undo_draws(self, log):
        """The path that drew `log` did not run: queue its tensors, in order, for the path that
        runs instead.  Both draw the same shapes in the same order (the reference's), so the
        generator is consumed once and the results are those of the reference's draw order."""
        if log:
            self._noise_pool = list(log) + list(self._noise_pool or [])

    def _finish(self, device):
        # the status bits are OR-accumulated on the device: a caller that runs several
        # ops (kalman_elbo) defers the one blocking read to its end
        if self.check_numerics and not self._defer_status:
            F.raise_on_status(device)

    # -- filter (state_space_model.py:172-373) ---------------------------------
    def kalman_filter(
        self,
        as_: torch.Tensor,
        sample_control: SampleControl,
        observation_mask: Optional[torch.Tensor] = None,
        learn_weight_model: bool = True,
        symmetrize_covariance: bool = True,
        burn_in: int = 0,
    ):
        T, B = as_.shape[:2]
        state_noise, obs_noise = self._modes(sample_control, filtering=True)
        self.weight_model.clear_hidden_state()
        if observation_mask is None:
            out = self._filter_unmasked(as_, state_noise, obs_noise, learn_weight_model,
                                        symmetrize_covariance, burn_in)
        else:
            out = self._filter_masked(as_, observation_mask.to(as_.dtype), state_noise, obs_noise,
                                      learn_weight_model, symmetrize_covariance, burn_in)
        self._finish(as_.device)
        return out

    def _filter_unmasked(self, as_, state_noise, obs_noise, learn_weight_model, symmetrize, burn_in):
        T, B = as_.shape[:2]
        if self.rng_parity:
            # the reference draws the imputation noise every step although its
            # result is unused without a mask; consume the generator alike
            for _ in range(T):
                if state_noise:
                    self._draw((B, self.z_dim), as_)
                if obs_noise:
                    self._draw((B, self.a_dim), as_)
        # row 0: uniform weights; rows 1..T: weights after seeing a_0..a_{T-1}
        alpha = self.weight_model(as_, lead=True)
        if not learn_weight_model:
            alpha = alpha.detach()
        m, P, mn, Pn, As, Cs = F.kalman_filter_steps(
            as_, alpha, None, self.mat_A_K, self.mat_C_K, self.mat_Q, self.mat_R,
            self.initial_state_mean, self.initial_state_covariance,
            symmetrize=symmetrize, burn_in=burn_in)
        return m, P, mn, Pn, As, Cs, as_, alpha[:T]

    def _filter_masked(self, as_, mask, state_noise, obs_noise, learn_weight_model, symmetrize,
                       burn_in):
        T, B = as_.shape[:2]
        A_K, C_K, Q, R = self.mat_A_K, self.mat_C_K, self.mat_Q, self.mat_R
        L_Q = F.cholesky_factor(Q) if state_noise else None
        L_R = F.cholesky_factor(R) if obs_noise else None
        def matvec(M, v):
            # per-sequence M v; torch.bmm on 65536 4x4 problems is another ~0.9 ms library call
            return (M * v.unsqueeze(-2)).sum(-1)

        def mix(w, bases):
            # sum_k w[b,k] bases[k] as K broadcast multiply-adds: as a (B,K)x(K,n*n)
            # product rocBLAS takes ~0.9 ms at B = 65536 (K = 3 is a degenerate GEMM)
            flat = bases.reshape(bases.shape[0], 1, -1)
            out = w[:, 0:1] * flat[0]
            for k in range(1, bases.shape[0]):
                out = torch.addcmul(out, w[:, k:k + 1], flat[k])
            return out.view(w.shape[0], *bases.shape[1:])
        w = torch.full((B, self.K), 1.0 / self.K, dtype=as_.dtype, device=as_.device)
        m_pred = self.initial_state_mean.expand(B, -1).contiguous()
        P_pred = self.initial_state_covariance.expand(B, -1, -1).contiguous()
        layers = getattr(self.weight_model, "layers", None)
        if (torch.is_grad_enabled() and as_.is_cuda and self.in_kernel_weight_network
                and layers is not None and len(layers) == 2
                and F.masked_mlp_grad_supported(layers[0].out_features, self.z_dim, self.a_dim,
                                                self.K, as_.dtype)):
            # two-layer perceptron under autograd: one launch per direction
            with self._recording_draws() as log:
                out = F.masked_filter_mlp_grad(
                    as_, mask, layers[0], layers[1], A_K, C_K, Q, R, self.initial_state_mean,
                    self.initial_state_covariance,
                    *self._filter_draws_split(as_, state_noise, obs_noise, L_Q, L_R),
                    symmetrize=symmetrize, burn_in=burn_in, learn_weight_model=learn_weight_model)
            if out is not None:
                return out[:7] + (out[7][:T],)
            self._undo_draws(log)
        lstm, head = getattr(self.weight_model, "lstm", None), getattr(self.weight_model, "linear", None)
        if (torch.is_grad_enabled() and as_.is_cuda and lstm is not None and head is not None
                and getattr(self.weight_model, "use_hip_recurrence", False)
                and self.weight_model.hidden is None):
            # the default LSTM network under autograd: one node, the steps walked by hand in
            # both directions
            with self._recording_draws() as log:
                out = masked_lstm.masked_filter_lstm_grad(
                    as_, mask, lstm, head, A_K, C_K, Q, R, self.initial_state_mean,
                    self.initial_state_covariance,
                    *self._filter_draws_split(as_, state_noise, obs_noise, L_Q, L_R),
                    symmetrize=symmetrize, burn_in=burn_in, learn_weight_model=learn_weight_model)
            if out is not None:
                self.weight_model.hidden = (out[8], out[9])
                self.weight_model.count += T
                return out[:7] + (out[7][:T],)
            self._undo_draws(log)
        if (not torch.is_grad_enabled() and as_.is_cuda
                and F.impute_supported(self.z_dim, self.a_dim, self.K, as_.dtype, as_.device)):
            layers = getattr(self.weight_model, "layers", None)
            if (layers is not None and len(layers) == 2 and self.in_kernel_weight_network
                    and F.masked_mlp_fits(layers[0].out_features, self.a_dim, self.K, as_.dtype)):
                # two-layer perceptron: the network runs inside the filter kernel
                with self._recording_draws() as log:
                    fast = self._filter_masked_mlp(as_, mask, state_noise, obs_noise, symmetrize,
                                                   L_Q, L_R, layers)
                if fast is not None:
                    return fast
                self._undo_draws(log)
            with self._recording_draws() as log:
                fast = self._filter_masked_inference(as_, mask, state_noise, obs_noise,
                                                     symmetrize, w, m_pred, P_pred, L_Q, L_R)
            if fast is not None:
                return fast
            self._undo_draws(log)
        A_t = mix(w, A_K)
        out = {k: [] for k in ("m", "P", "mn", "Pn", "A", "C", "aw", "w")}
        for t in range(T):
            C_t = mix(w, C_K)
            # imputation (:232-277); m_pred already is the one-step prediction
            z_imp = matvec(A_t, m_pred)
            if state_noise:
                z_imp = z_imp + self._draw((B, self.z_dim), as_) @ L_Q.T
            a_imp = matvec(C_t, z_imp)
            if obs_noise:
                a_imp = a_imp + self._draw((B, self.a_dim), as_) @ L_R.T
            mu = mask[t].unsqueeze(-1)
            a_w = mu * as_[t] + (1.0 - mu) * a_imp
            w_next = self.weight_model(a_w.unsqueeze(0)).squeeze(0)
            if not learn_weight_model:
                w_next = w_next.detach()
            A_next = mix(w_next, A_K)
            m, P, mn, Pn, _, _ = F.kalman_filter_steps(
                as_[t:t + 1], torch.stack([w, w_next]), mask[t:t + 1], A_K, C_K, Q, R,
                m_pred, P_pred, symmetrize=symmetrize, burn_in=1 if t < burn_in else 0,
                want_mats=False)
            m_pred, P_pred = mn[0], Pn[0]
            for k, v in zip(("m", "P", "mn", "Pn", "A", "C", "aw", "w"),
                            (m[0], P[0], m_pred, P_pred, A_t, C_t, a_w, w)):
                out[k].append(v)
            w, A_t = w_next, A_next
        out["A"].append(A_t)
        out["C"].append(mix(w, C_K))
        st = {k: torch.stack(v) for k, v in out.items()}
        return st["m"], st["P"], st["mn"], st["Pn"], st["A"], st["C"], st["aw"], st["w"]

    def _filter_masked_mlp(self, as_, mask, state_noise, obs_noise, symmetrize, L_Q, L_R, layers):
        """The masked filter when no gradient is recorded and the weight network is a
        two-layer perceptron: ONE launch for the whole sequence, the network evaluated
        inside the recursion.  The noise is drawn in the order of the step-by-step path
        (per step: state, then observation).  None when no kernel covers the shapes."""
        eps_z, eps_a = self._filter_draws(as_, state_noise, obs_noise)
        out = F.masked_filter_mlp(
            as_, mask, F.pack_mlp2(layers[0], layers[1]).to(as_.dtype), layers[0].out_features,
            self.mat_A_K, self.mat_C_K, self.mat_Q, self.mat_R, self.initial_state_mean,
            self.initial_state_covariance,
            eps_z, L_Q, eps_a, L_R, symmetrize=symmetrize)
        if out is None:
            return None
        m, P, mn, Pn, As, Cs, a_w, alpha = out
        return m, P, mn, Pn, As, Cs, a_w, alpha[:as_.shape[0]]

    def _filter_masked_inference(self, as_, mask, state_noise, obs_noise, symmetrize, w, m_pred,
                                 P_pred, L_Q, L_R):
        """The masked filter when no gradient is recorded (the reference's imputation /
        evaluation use): per step ONE launch imputes the observation the weight network
        sees, the network steps from its carried state, and the one-step filter kernel
        also writes the mixed A_t / C_t it used.  None when the imputation kernel is not
        compiled for these dimensions."""
        T, B = as_.shape[:2]
        A_K, C_K, Q, R = self.mat_A_K, self.mat_C_K, self.mat_Q, self.mat_R
        make_stepper = getattr(self.weight_model, "inference_stepper", None)
        if make_stepper is not None and F.embedding_dims(
                self.z_dim, self.a_dim, self.K, as_.dtype) == (self.z_dim, self.a_dim, self.K):
            stepper = make_stepper(as_[0])
            if stepper is not None:
                return self._filter_masked_stepper(as_, mask, state_noise, obs_noise, symmetrize,
                                                   w, m_pred, P_pred, L_Q, L_R, stepper)
        out = {k: [] for k in ("m", "P", "mn", "Pn", "A", "C", "aw", "w")}
        As2 = Cs2 = None
        for t in range(T):
            eps_z = self._draw((B, self.z_dim), as_) if state_noise else None
            eps_a = self._draw((B, self.a_dim), as_) if obs_noise else None
            a_w = F.impute_observation(w, A_K, C_K, m_pred, as_[t], mask[t], eps_z, L_Q, eps_a, L_R)
            if a_w is None:
                return None
            w_next = self.weight_model(a_w.unsqueeze(0)).squeeze(0)
            m, P, mn, Pn, As2, Cs2 = F.kalman_filter_steps(
                as_[t:t + 1], torch.stack([w, w_next]), mask[t:t + 1], A_K, C_K, Q, R,
                m_pred, P_pred, symmetrize=symmetrize, burn_in=0, want_mats=True)
            m_pred, P_pred = mn[0], Pn[0]
            for k, v in zip(("m", "P", "mn", "Pn", "A", "C", "aw", "w"),
                            (m[0], P[0], m_pred, P_pred, As2[0], Cs2[0], a_w, w)):
                out[k].append(v)
            w = w_next
        out["A"].append(As2[1])
        out["C"].append(Cs2[1])
        st = {k: torch.stack(v) for k, v in out.items()}
        return st["m"], st["P"], st["mn"], st["Pn"], st["A"], st["C"], st["aw"], st["w"]

    def _filter_masked_stepper(self, as_, mask, state_noise, obs_noise, symmetrize, w, m_pred,
                               P_pred, L_Q, L_R, stepper):
        """_filter_masked_inference with a weight network that steps into the caller's
        tensors: every per-step result is written into its slice of the (T, ...) tensor
        that is returned (three network launches, the imputation and the one-step filter
        per time step; no per-step tensors, no stacking)."""
        T, B = as_.shape[:2]
        n, p, K = self.z_dim, self.a_dim, self.K
        new = lambda *s: torch.empty(s, dtype=as_.dtype, device=as_.device)
        c = lambda t: t.contiguous()
        A_K, C_K, Q, R = c(self.mat_A_K), c(self.mat_C_K), c(self.mat_Q), c(self.mat_R)
        as_, mask = c(as_), c(mask)
        m, P, mn, Pn = new(T, B, n), new(T, B, n, n), new(T, B, n), new(T, B, n, n)
        As, Cs, aw, ws = new(T + 1, B, n, n), new(T + 1, B, p, n), new(T, B, p), new(T + 1, B, K)
        ws[0].copy_(w)
        for t in range(T):
            eps_z = self._draw((B, n), as_) if state_noise else None
            eps_a = self._draw((B, p), as_) if obs_noise else None
            F.impute_observation(ws[t], A_K, C_K, m_pred, as_[t], mask[t], eps_z, L_Q, eps_a, L_R,
                                 out=aw[t])
            stepper.step(aw[t], ws[t + 1])
            # writes A_t, C_t and A_{t+1}, C_{t+1}; the next step writes the latter again
            F.filter_step_into(as_[t:t + 1], ws[t:t + 2], mask[t:t + 1], A_K, C_K, Q, R, m_pred,
                               P_pred, m[t:t + 1], P[t:t + 1], mn[t:t + 1], Pn[t:t + 1],
                               As[t:t + 2], Cs[t:t + 2], symmetrize=symmetrize)
            m_pred, P_pred = mn[t], Pn[t]
        self.weight_model.finish_steps(stepper, T)
        return m, P, mn, Pn, As, Cs, aw, ws[:T]

    def _filter_draws(self, as_, state_noise, obs_noise):
        """(eps_z (T,B,n), eps_a (T,B,p)) of the masked filter's imputation, drawn in the
        order of the step-by-step path (per step: state, then observation); None where
        the mode is "mean"."""
        T, B = as_.shape[:2]
        if not (state_noise or obs_noise):
            return None, None
        ez, ea = [], []
        for _ in range(T):
            if state_noise:
                ez.append(self._draw((B, self.z_dim), as_))
            if obs_noise:
                ea.append(self._draw((B, self.a_dim), as_))
        return (torch.stack(ez) if state_noise else None, torch.stack(ea) if obs_noise else None)

    def _filter_draws_split(self, as_, state_noise, obs_noise, L_Q, L_R):
        """(eps_z, L_Q, eps_a, L_R) in the argument order of the masked-filter kernels."""
        eps_z, eps_a = self._filter_draws(as_, state_noise, obs_noise)
        return eps_z, L_Q, eps_a, L_R

    def _smoother_draws(self, as_, state_noise, obs_noise):
        """(eps_state (T,B,n), eps_obs (T,B,p)) of the smoother's samples; None where the
        mode is "mean"."""
        T, B = as_.shape[:2]
        eps_state = eps_obs = None
        if self.rng_parity and (state_noise or obs_noise):
            # draw order of the reference: t = T-1 .. 0, state then observation
            es, eo = [None] * T, [None] * T
            for t in reversed(range(T)):
                if state_noise:
                    es[t] = self._draw((B, self.z_dim), as_)
                if obs_noise:
                    eo[t] = self._draw((B, self.a_dim), as_)
            eps_state = torch.stack(es) if state_noise else None
            eps_obs = torch.stack(eo) if obs_noise else None
        else:
            if state_noise:
                eps_state = self._draw((T, B, self.z_dim), as_)
            if obs_noise:
                eps_obs = self._draw((T, B, self.a_dim), as_)
        return eps_state, eps_obs

    @torch.no_grad()
    def impute(self, as_, sample_control: SampleControl, observation_mask=None,
               symmetrize_covariance: bool = True):
        """``(filter_as, as_resampled)``: the two tensors an imputation consumer reads from
        ``kalman_filter`` followed by ``kalman_smooth`` (kvae/kalman_vae.py:141-160 as
        kvae/evaluation.py:100-140 uses it) - the observations with the masked steps filled
        in by the filter, and the observations re-generated from the smoothed states.
        With a two-layer perceptron weight network (or without a mask) the filter, the
        network and the smoother are ONE launch that writes nothing else; otherwise the two
        calls run.  Values only."""
        T, B = as_.shape[:2]
        f_state, f_obs = self._modes(sample_control, filtering=True)
        s_state, s_obs = self._modes(sample_control, filtering=False)
        mask = None if observation_mask is None else observation_mask.to(as_.dtype)
        layers = getattr(self.weight_model, "layers", None)
        in_kernel = (layers is not None and len(layers) == 2 and self.in_kernel_weight_network
                     and F.masked_mlp_fits(layers[0].out_features, self.a_dim, self.K,
                                           as_.dtype))
        if (as_.is_cuda and (mask is None or in_kernel)
                and F.rollout_supported(self.z_dim, self.a_dim, self.K, as_.dtype)):
            self.weight_model.clear_hidden_state()
            with self._recording_draws() as log, self.cached_matrices():
                common = (self.mat_A_K, self.mat_C_K, self.mat_Q, self.mat_R,
                          self.initial_state_mean, self.initial_state_covariance)
                if mask is None:
                    if self.rng_parity:
                        self._filter_draws(as_, f_state, f_obs)     # drawn and unused, as the reference
                    alpha = self.weight_model(as_, lead=True)
                    eps_state, eps_obs = self._smoother_draws(as_, s_state, s_obs)
                    out = F.filter_smooth_fused(
                        as_, None, *common, alpha=alpha,
                        L_R=self._cholesky_R() if s_obs else None, eps_state=eps_state,
                        eps_obs=eps_obs, symmetrize=symmetrize_covariance,
                        want=("as_resampled",))
                    filter_as = as_
                else:
                    eps_z, eps_a = self._filter_draws(as_, f_state, f_obs)
                    eps_state, eps_obs = self._smoother_draws(as_, s_state, s_obs)
                    out = F.filter_smooth_fused(
                        as_, mask, *common,
                        packed=F.pack_mlp2(layers[0], layers[1]).to(as_.dtype),
                        hidden=layers[0].out_features, eps_z=eps_z,
                        L_Q=F.cholesky_factor(self.mat_Q) if f_state else None, eps_a=eps_a,
                        L_R=self._cholesky_R() if (f_obs or s_obs) else None,
                        eps_state=eps_state, eps_obs=eps_obs, symmetrize=symmetrize_covariance,
                        want=("a_w", "as_resampled"))
                    filter_as = out["a_w"] if out is not None else None
            if out is not None:
                self._finish(as_.device)
                return filter_as, out["as_resampled"]
            # no kernel for this call after all: the two calls below are served the noise
            # already drawn, in the order they ask for it
            self._undo_draws(log)
        out = self.kalman_filter(as_, sample_control, observation_mask=observation_mask,
                                 symmetrize_covariance=symmetrize_covariance)
        smoothed = self.kalman_smooth(as_, *out[:6], sample_control=sample_control,
                                      symmetrize_covariance=symmetrize_covariance)
        return out[6], smoothed[3]

    # -- smoother (state_space_model.py:375-484) --------------------------------
    def kalman_smooth(
        self,
        as_: torch.Tensor,
        filter_means: torch.Tensor,
        filter_covariances: torch.Tensor,
        filter_next_means: torch.Tensor,
        filter_next_covariances: torch.Tensor,
        mat_As: torch.Tensor,
        mat_Cs: torch.Tensor,
        sample_control: SampleControl,
        symmetrize_covariance: bool = True,
        burn_in: int = 0,
    ):
        T, B, _ = as_.shape
        state_noise, obs_noise = self._modes(sample_control, filtering=False)
        eps_state, eps_obs = self._smoother_draws(as_, state_noise, obs_noise)
        L_R = self._cholesky_R() if obs_noise else None
        out = F.kalman_smooth_steps(
            filter_means, filter_covariances, filter_next_means, filter_next_covariances,
            mat_As, mat_Cs, LR=L_R, eps_state=eps_state, eps_obs=eps_obs,
            symmetrize=symmetrize_covariance, burn_in=burn_in)
        self._finish(as_.device)
        return out

    # -- open-loop rollout (state_space_model.py:486-575) -------------------------
    def predict_future(self, as_, means, covariances, next_means, next_covariances, mat_As,
                       mat_Cs, num_steps, sample_control: SampleControl):
        """Roll the filter ``num_steps`` steps past the end of ``as_`` without
        observations.  Same returns as the reference: ``(as_tensor, means,
        covariances, next_means, next_covariances, mat_As_list, mat_Cs_list)`` with
        lists of per-step tensors.  As in the reference the weight network is fed the
        whole grown sequence at every step and keeps its cached recurrent state
        between those calls (:544; it is not cleared), and with "sample" the next
        covariance is reset to Q (:550-551).  When no autograd graph is recorded the
        rollout runs in the HIP library (_predict_future_kernels); otherwise as plain
        torch ops around the weight network."""
        T, B = as_.shape[:2]
        state_noise = wants_noise(sample_control.state_transition, "state_transition")
        if (as_.is_cuda and not torch.is_grad_enabled() and num_steps > 0
                and F.rollout_supported(self.z_dim, self.a_dim, self.K, as_.dtype)):
            return self._predict_future_kernels(as_, means, covariances, next_means,
                                                next_covariances, mat_As, mat_Cs, num_steps,
                                                state_noise)
        as_list = list(as_)
        mat_As_list, mat_Cs_list = list(mat_As), list(mat_Cs)
        means, covariances = list(means), list(covariances)
        next_means, next_covariances = list(next_means), list(next_covariances)
        state_noise = wants_noise(sample_control.state_transition, "state_transition")
        as_tensor = torch.stack(as_list, dim=0)
        for t_future in range(num_steps):
            t = T + t_future
            means.append(next_means[-1])
            covariances.append(next_covariances[-1])
            next_z = next_means[-1].reshape(-1, self.z_dim)
            if state_noise:
                L = torch.linalg.cholesky(next_covariances[-1])
                eps = self._draw(next_z.shape, next_z)
                next_z = next_z + (L @ eps.unsqueeze(-1)).squeeze(-1)
                next_z = next_z.detach()          # the reference uses .sample(), not .rsample()
            next_a = (mat_Cs_list[t] * next_z.unsqueeze(-2)).sum(-1)
            as_list.append(next_a.view(B, self.a_dim))
            as_tensor = torch.stack(as_list, dim=0)
            weights = self.weight_model(as_tensor)
            mat_A = torch.einsum("bk,kij->bij", weights[-1], self.mat_A_K)
            mat_C = torch.einsum("bk,kij->bij", weights[-1], self.mat_C_K)
            mat_As_list.append(mat_A)
            mat_Cs_list.append(mat_C)
            next_mean = (mat_A * next_means[-1].unsqueeze(-2)).sum(-1)
            if state_noise:
                next_covariance = self.mat_Q.unsqueeze(0).repeat(B, 1, 1)
            else:
                next_covariance = mat_A @ next_covariances[-1] @ mat_A.transpose(1, 2) + self.mat_Q
            next_covariance = (next_covariance + next_covariance.transpose(1, 2)) / 2.0
            next_means.append(next_mean)
            next_covariances.append(next_covariance)
        return (as_tensor, means, covariances, next_means, next_covariances, mat_As_list,
                mat_Cs_list)

    def _predict_future_kernels(self, as_, means, covariances, next_means, next_covariances,
                                mat_As, mat_Cs, num_steps, state_noise):
        """predict_future through kvae_predict_future.  Two-layer perceptron: all new
        steps are one launch (the network runs in the kernel).  Any other network: per
        new step one launch emits the observation, the network runs over the grown
        sequence from its carried state exactly as the reference does, and one launch
        advances the prediction with the new weights."""
        T, B = as_.shape[:2]
        S = int(num_steps)
        A_K, C_K, Q = self.mat_A_K, self.mat_C_K, self.mat_Q
        mn, Pn, C_last = next_means[-1], next_covariances[-1], mat_Cs[T]
        layers = getattr(self.weight_model, "layers", None)
        in_kernel = (layers is not None and len(layers) == 2 and self.in_kernel_weight_network
                     and F.masked_mlp_fits(layers[0].out_features, self.a_dim, self.K,
                                           as_.dtype))
        eps = None
        if in_kernel:
            if state_noise:
                eps = torch.stack([self._draw((B, self.z_dim), as_) for _ in range(S)])
            a_new, m_new, P_new, mn_new, Pn_new, As_new, Cs_new = F.predict_rollout(
                "all", A_K, C_K, Q, mn, Pn, C_last, eps,
                F.pack_mlp2(layers[0], layers[1]).to(as_.dtype), layers[0].out_features,
                sample=state_noise, steps=S)
            as_tensor = torch.cat([as_, a_new], dim=0)
        else:
            rows = {k: [] for k in ("m", "P", "mn", "Pn", "A", "C")}
            as_tensor = as_
            for _ in range(S):
                eps = self._draw((B, self.z_dim), as_).unsqueeze(0) if state_noise else None
                a_s, m_s, P_s = F.predict_rollout("emit", A_K, C_K, Q, mn, Pn, C_last, eps,
                                                  sample=state_noise)
                as_tensor = torch.cat([as_tensor, a_s], dim=0)
                weights = self.weight_model(as_tensor)
                mn_s, Pn_s, A_s, C_s = F.predict_rollout("advance", A_K, C_K, Q, mn, Pn,
                                                         alpha=weights[-1], sample=state_noise)
                mn, Pn, C_last = mn_s[0], Pn_s[0], C_s[0]
                for k, v in zip(("m", "P", "mn", "Pn", "A", "C"),
                                (m_s, P_s, mn_s, Pn_s, A_s, C_s)):
                    rows[k].append(v[0])
            m_new, P_new, mn_new, Pn_new, As_new, Cs_new = (
                torch.stack(rows[k]) for k in ("m", "P", "mn", "Pn", "A", "C"))
        self._finish(as_.device)
        return (as_tensor, list(means) + list(m_new), list(covariances) + list(P_new),
                list(next_means) + list(mn_new), list(next_covariances) + list(Pn_new),
                list(mat_As) + list(As_new), list(mat_Cs) + list(Cs_new))
