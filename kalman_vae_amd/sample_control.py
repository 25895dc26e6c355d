"""Sampling switches of the Kalman VAE (same fields and defaults as the
reference's kvae/sample_control.py:4-8)."""
from typing import NamedTuple

_MODES = ("sample", "mean")


class SampleControl(NamedTuple):
    encoder: str = "sample"
    decoder: str = "mean"
    state_transition: str = "sample"
    observation: str = "sample"


def wants_noise(mode: str, field: str) -> bool:
    """True for "sample", False for "mean"; anything else raises the same
    ValueError text the reference raises (state_space_model.py:244-249)."""
    if mode == "sample":
        return True
    if mode == "mean":
        return False
    raise ValueError("Invalid sample_control.{}: {}".format(field, mode))
