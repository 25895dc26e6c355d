"""The full Kalman VAE behind the reference's interface
(kvae/kalman_vae.py:16-292): frame encoder -> Kalman filter / smoother / ELBO
terms -> frame decoder, one scalar objective.

Everything between the encoded observations ``as_`` and the three Kalman terms
is ``kalman_elbo`` (HIP kernels); the encoder, the decoder and their two terms
are plain PyTorch, added to the objective with the reference's weights and
signs (kalman_vae.py:71-90, 104-127, 168-187, 262-265).
"""
import logging
from typing import Optional

import torch
import torch.distributions as D
import torch.nn as nn
from torch.distributions.utils import _standard_normal

from .elbo import kalman_elbo
from .misc import _validate_shape, aggregate
from .sample_control import SampleControl
from .state_space_model import StateSpaceModel
from .variational_autoencoder import BernoulliDecoder, Encoder, GaussianDecoder

logger = logging.getLogger(__name__)

_DECODERS = {"gaussian": GaussianDecoder, "bernoulli": BernoulliDecoder}


class KalmanVariationalAutoencoder(nn.Module):
    def __init__(self, image_size, image_channels, a_dim, z_dim, K,
                 init_transition_reg_weight: float, init_observation_reg_weight: float,
                 learn_noise_covariance: bool, init_noise_scale: float,
                 dynamics_parameter_network: str, decoder_type: str = "gaussian"):
        super().__init__()
        if decoder_type not in _DECODERS:
            raise ValueError("Unknown decoder type: {}".format(decoder_type))
        self.encoder = Encoder(image_size, image_channels, a_dim)
        self.decoder = _DECODERS[decoder_type](a_dim, image_size, image_channels)
        self.state_space_model = StateSpaceModel(
            a_dim=a_dim, z_dim=z_dim, K=K,
            dynamics_parameter_network=dynamics_parameter_network,
            init_transition_reg_weight=init_transition_reg_weight,
            init_observation_reg_weight=init_observation_reg_weight,
            learn_noise_covariance=learn_noise_covariance,
            init_noise_scale=init_noise_scale)
        self.a_dim = a_dim
        self.z_dim = z_dim
        self.register_buffer("_zero_val", torch.tensor(0.0))

    def _encode(self, xs, sample_control):
        """q(a|x) over all frames and the encoded sequence the reference would use
        (kalman_vae.py:75-94): one standard-normal draw when sampling."""
        T, B = xs.shape[:2]
        q = self.encoder(xs.reshape(T * B, *xs.shape[2:]))
        if sample_control.encoder == "sample":
            eps = _standard_normal(q.loc.shape, dtype=q.loc.dtype, device=q.loc.device)
            a = q.loc + q.scale * eps
        elif sample_control.encoder == "mean" and not self.training:
            a = q.loc
        else:
            raise ValueError("Invalid sample control for encoder: {}".format(sample_control.encoder))
        return q, a.view(T, B, self.a_dim)

    def elbo(self, sample_control: SampleControl, xs: Optional[torch.Tensor] = None,
             as_: Optional[torch.Tensor] = None, observation_mask=None, reconst_weight=0.3,
             regularization_weight=1.0, kalman_weight=1.0, kl_weight=0.0, learn_weight_model=True,
             symmetrize_covariance=True, burn_in=0, sequence_operation="mean",
             batch_operation="mean"):
        if (xs is None) == (as_ is None):
            raise ValueError("Either as_ or xs must be provided" if xs is None
                             else "Only one of as_ and xs must be provided")
        agg = dict(sequence_operation=sequence_operation, batch_operation=batch_operation)
        q = None
        if xs is not None:
            T, B = xs.shape[:2]
            q, as_ = self._encode(xs, sample_control)
        else:
            T, B = as_.shape[:2]
            _validate_shape(as_, (T, B, self.a_dim), "as_")
            reconst_weight = regularization_weight = kl_weight = 0.0
            logger.info("reconst_weight = regularization_weight = kl_weight = 0.0")

        reconstruction = regularization = 0.0
        if xs is not None:
            frames = xs.reshape(T * B, *xs.shape[2:])
            flat_a = as_.view(T * B, self.a_dim)
            # ln p(x|a) summed over the pixels, and ln q(a|x) summed over a_dim
            log_px = self.decoder(flat_a).log_prob(frames).view(T, B, -1).sum(-1)
            log_qa = q.log_prob(flat_a).view(T, B, self.a_dim).sum(-1)
            reconstruction = reconst_weight * aggregate(log_px, sequence_length=T, batch_size=B, **agg)
            regularization = -regularization_weight * aggregate(log_qa, sequence_length=T,
                                                                batch_size=B, **agg)

        kalman, info = kalman_elbo(
            self.state_space_model, as_, sample_control, observation_mask=observation_mask,
            kalman_weight=kalman_weight, learn_weight_model=learn_weight_model,
            symmetrize_covariance=symmetrize_covariance, burn_in=burn_in, **agg)

        kl = self._zero_val
        if kl_weight != 0.0:
            prior = D.Normal(torch.zeros(self.a_dim, dtype=xs.dtype, device=xs.device),
                             torch.ones(self.a_dim, dtype=xs.dtype, device=xs.device))
            per_frame = D.kl.kl_divergence(q, prior).view(T, B, self.a_dim).sum(-1)
            kl = -aggregate(per_frame, sequence_length=T, batch_size=B, **agg)
        weighted_kl = kl_weight * kl

        objective = weighted_kl + kalman
        if xs is not None:
            objective = objective + reconstruction + regularization
        info.pop("elbo_terms", None)
        info.update({
            "reconst_weight": reconst_weight,
            "regularization_weight": regularization_weight,
            "kl_weight": kl_weight,
            "reconstruction": reconstruction,
            "regularization": regularization,
            "kl": weighted_kl,
        })
        return objective, info
