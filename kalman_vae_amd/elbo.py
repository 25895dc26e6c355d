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
        eps = _standard_normal((T, B, ssm.z_dim), dtype=as_.dtype, device=as_.device)
        obs, trans, post, _ = F.kalman_elbo_terms(
            as_, means, covariances, mat_As, mat_Cs, ssm.mat_Q, ssm.mat_R,
            ssm.initial_state_mean, ssm.initial_state_covariance, eps)
    if ssm.check_numerics:
        F.raise_on_status(as_.device)
    # aggregate(v) (kvae/misc.py:21-41) is a constant times the total of v; the three
    # totals are one deterministic tree sum on the HIP library, the weights
    # (kalman_weight, and the minus sign of the posterior term, kalman_vae.py:244-260)
    # one multiplication by a constant vector
    c = kalman_weight * aggregate_weights(T, B, sequence_operation, batch_operation)
    weighted = F.term_totals(obs, trans, post) * F.constant_vector((c, c, -c), obs)
    w_obs, w_trans, w_post = weighted[0], weighted[1], weighted[2]
    objective = weighted.sum()
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
