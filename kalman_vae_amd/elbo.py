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
):
    """Returns (objective, info) with the same ``info`` keys the reference
    fills for the Kalman terms and the filter/smoother tensors
    (kalman_vae.py:267-292)."""
    ssm = state_space_model
    T, B = as_.shape[:2]
    _validate_shape(as_, (T, B, ssm.a_dim), "as_")
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
    try:
        return _kalman_elbo(ssm, as_, sample_control, observation_mask, kalman_weight,
                            learn_weight_model, symmetrize_covariance, burn_in,
                            sequence_operation, batch_operation, eps)
    finally:
        ssm._noise_pool = None


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
