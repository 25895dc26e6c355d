"""The Kalman part of ``KalmanVariationalAutoencoder.elbo`` when it is called
with encoded observations ``as_`` (kvae/kalman_vae.py:92-101, 141-168,
191-260): filter -> smoother -> sample z ~ N(smoothed) -> three log-density
terms -> aggregate -> kalman_weight * (obs + trans - post).

The encoder/decoder terms of the full model are ordinary PyTorch-ROCm convs
and are not part of this package's hot path; add them to the returned
objective exactly as the reference does (kalman_vae.py:262-265).
"""
import torch
from torch.distributions.utils import _standard_normal

from . import functions as F
from .misc import _validate_shape, aggregate_weights
from .sample_control import SampleControl


def kalman_elbo(
    state_space_model,
    as_: torch.Tensor,
    sample_control: SampleControl,
    observation_mask=None,
    kalman_weight: float = 1.0,
    learn_weight_model: bool = True,
    symmetrize_covariance: bool = True,
    burn_in: int = 0,
    sequence_operation: str = "mean",
    batch_operation: str = "mean",
    fused=None,
):
    """Returns (objective, info) with the same ``info`` keys the reference
    fills for the Kalman terms and the filter/smoother tensors
    (kalman_vae.py:267-292).

    ``fused`` (default: ``state_space_model.fused_loss``): compute the objective with
    the one-launch-per-direction kernels, which never materialise the filter / smoother
    tensors; their ``info`` entries are then ``None``.  A training loop reads only the
    objective and the three scalar terms (examples/train.py:115-131 of the reference).
    Ignored (the separate kernels run) with an observation mask or for dimensions
    without a fused kernel."""
    ssm = state_space_model
    T, B = as_.shape[:2]
    _validate_shape(as_, (T, B, ssm.a_dim), "as_")
    if fused is None:
        fused = ssm.fused_loss
    if (fused and observation_mask is None and as_.is_cuda
            and F.loss_supported(ssm.z_dim, ssm.a_dim, ssm.K, as_.dtype)):
        return _kalman_elbo_fused(ssm, as_, sample_control, kalman_weight, learn_weight_model,
                                  symmetrize_covariance, burn_in, sequence_operation,
                                  batch_operation)
    eps = None
    if not ssm.rng_parity and observation_mask is None:
        # free draw order: all the noise of the step (smoother state and observation
        # noise, ELBO sample) in ONE launch, handed out as contiguous pieces
        st, ob = ssm._modes(sample_control, filtering=False)
        shapes = ([(T, B, ssm.z_dim)] if st else []) + ([(T, B, ssm.a_dim)] if ob else [])
        shapes.append((T, B, ssm.z_dim))
        sizes = [s[0] * s[1] * s[2] for s in shapes]
        flat = _standard_normal((sum(sizes),), dtype=as_.dtype, device=as_.device)
        pieces = [piece.view(shape) for piece, shape in zip(flat.split(sizes), shapes)]
        ssm._noise_pool, eps = pieces[:-1], pieces[-1]
    ssm._defer_status = True       # one blocking read of the status word, at the end
    try:
        return _kalman_elbo(ssm, as_, sample_control, observation_mask, kalman_weight,
                            learn_weight_model, symmetrize_covariance, burn_in,
                            sequence_operation, batch_operation, eps)
    finally:
        ssm._noise_pool = None
        ssm._defer_status = False


def _kalman_elbo_fused(ssm, as_, sample_control, kalman_weight, learn_weight_model,
                       symmetrize_covariance, burn_in, sequence_operation, batch_operation):
    """kalman_vae.py:141-260 through kvae_kalman_loss_forward / _backward."""
    T, B = as_.shape[:2]
    n, p = ssm.z_dim, ssm.a_dim
    f_state, f_obs = ssm._modes(sample_control, filtering=True)     # same errors as the filter
    s_state, s_obs = ssm._modes(sample_control, filtering=False)
    ssm.weight_model.clear_hidden_state()
    draw = lambda *shape: _standard_normal(shape, dtype=as_.dtype, device=as_.device)
    if ssm.rng_parity:
        # The reference draws the imputation noise of the filter and the samples of the
        # smoother although the objective never reads them: consume the generator alike
        # (state_space_model.py:232-235, 251-254, 393-472), then draw the ELBO sample.
        for _ in range(T):
            if f_state:
                draw(B, n)
            if f_obs:
                draw(B, p)
        for _ in range(T):
            if s_state:
                draw(B, n)
            if s_obs:
                draw(B, p)
    eps = draw(T, B, n)
    alpha = ssm.weight_model(as_, lead=True)
    if not learn_weight_model:
        alpha = alpha.detach()
    c = kalman_weight * aggregate_weights(T, B, sequence_operation, batch_operation)
    with ssm.cached_matrices():
        objective, weighted, terms = F.kalman_loss(
            as_, alpha, None, ssm.mat_A_K, ssm.mat_C_K, ssm.mat_Q, ssm.mat_R,
            ssm.initial_state_mean, ssm.initial_state_covariance, eps, c, c, -c,
            symmetrize=symmetrize_covariance, burn_in=burn_in, want_terms=ssm.fused_loss_terms)
    if ssm.check_numerics:
        F.raise_on_status(as_.device)
    info = {
        "kalman_weight": kalman_weight,
        "kalman_observation_log_likelihood": weighted[0],
        "kalman_state_transition_log_likelihood": weighted[1],
        "kalman_posterior_log_likelihood": weighted[2],
        "observation_mask": None,
        "weights": alpha[:T],
        "as": as_,
    }
    for key in ("filter_means", "filter_covariances", "filter_next_means",
                "filter_next_covariances", "mat_As", "mat_Cs", "means", "covariances", "zs",
                "filter_as", "as_resampled"):
        info[key] = None
    if ssm.fused_loss_terms:
        info["elbo_terms"] = (terms[0], terms[1], terms[2])
    return objective, info


def _kalman_elbo(ssm, as_, sample_control, observation_mask, kalman_weight, learn_weight_model,
                 symmetrize_covariance, burn_in, sequence_operation, batch_operation, eps):
    T, B = as_.shape[:2]
    with ssm.cached_matrices():     # Q, R and chol(R) are formed once for the whole step
        (filter_means, filter_covariances, filter_next_means, filter_next_covariances,
         mat_As, mat_Cs, filter_as, weights) = ssm.kalman_filter(
            as_, sample_control=sample_control, observation_mask=observation_mask,
            learn_weight_model=learn_weight_model, symmetrize_covariance=symmetrize_covariance,
            burn_in=burn_in)
        means, covariances, zs, as_resampled = ssm.kalman_smooth(
            as_, filter_means=filter_means, filter_covariances=filter_covariances,
            filter_next_means=filter_next_means, filter_next_covariances=filter_next_covariances,
            mat_As=mat_As, mat_Cs=mat_Cs, sample_control=sample_control,
            symmetrize_covariance=symmetrize_covariance, burn_in=burn_in)

        # zs_distrib.rsample(): one (T,B,n) standard-normal draw, always sampled
        if eps is None:
            eps = _standard_normal((T, B, ssm.z_dim), dtype=as_.dtype, device=as_.device)
        # aggregate(v) (kvae/misc.py:21-41) is a constant times the total of v: the
        # constant, kalman_weight and the minus sign of the posterior term
        # (kalman_vae.py:244-260) are the three weights of one fused node
        c = kalman_weight * aggregate_weights(T, B, sequence_operation, batch_operation)
        objective, weighted, terms, _ = F.kalman_elbo_objective(
            as_, means, covariances, mat_As, mat_Cs, ssm.mat_Q, ssm.mat_R,
            ssm.initial_state_mean, ssm.initial_state_covariance, eps, c, c, -c)
    if ssm.check_numerics:
        F.raise_on_status(as_.device)
    obs, trans, post = terms[0], terms[1], terms[2]
    w_obs, w_trans, w_post = weighted[0], weighted[1], weighted[2]
    info = {
        "kalman_weight": kalman_weight,
        "kalman_observation_log_likelihood": w_obs,
        "kalman_state_transition_log_likelihood": w_trans,
        "kalman_posterior_log_likelihood": w_post,
        "observation_mask": observation_mask,
        "filter_means": filter_means,
        "filter_covariances": filter_covariances,
        "filter_next_means": filter_next_means,
        "filter_next_covariances": filter_next_covariances,
        "mat_As": mat_As,
        "mat_Cs": mat_Cs,
        "weights": weights,
        "means": means,
        "covariances": covariances,
        "as": as_,
        "zs": zs,
        "filter_as": filter_as,
        "as_resampled": as_resampled,
        "elbo_terms": (obs, trans, post),
    }
    return objective, info
