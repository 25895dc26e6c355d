"""MI355X-native Kalman filter / RTS smoother / Kalman-ELBO hot path of the
Kalman VAE, behind the reference's ``StateSpaceModel`` interface.

Compute runs only in the HIP library kalman_vae_amd/csrc/libkvae_kalman.so
(gfx950); there is no CPU implementation in this package.
"""
from .sample_control import SampleControl
from .misc import aggregate
from .dynamics_parameter_network import LSTMModel, MLPModel
from .state_space_model import StateSpaceModel
from .elbo import kalman_elbo
from .variational_autoencoder import BernoulliDecoder, Encoder, GaussianDecoder
from .kalman_vae import KalmanVariationalAutoencoder
from .functions import (kalman_filter_steps, kalman_smooth_steps, kalman_elbo_terms,
                        read_status)

__all__ = ["SampleControl", "aggregate", "LSTMModel", "MLPModel", "StateSpaceModel",
           "kalman_elbo", "KalmanVariationalAutoencoder", "Encoder", "GaussianDecoder",
           "BernoulliDecoder", "kalman_filter_steps", "kalman_smooth_steps", "kalman_elbo_terms",
           "read_status"]
