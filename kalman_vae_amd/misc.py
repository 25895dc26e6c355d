"""Shape check and (T,B) -> scalar reduction used by the ELBO terms
(reference: kvae/misc.py:14-41)."""
from typing import Tuple

import torch


def _validate_shape(value: torch.Tensor, expected_shape: Tuple[int, ...], name: str):
    if tuple(value.shape) != tuple(expected_shape):
        raise ValueError(
            f"{name} must have shape {tuple(expected_shape)}, but got {value.shape}"
        )


_REDUCERS = {"mean": torch.mean, "sum": torch.sum}


def aggregate(value, sequence_length, batch_size, sequence_operation, batch_operation):
    """Reduce over the sequence axis first, then over the batch axis."""
    _validate_shape(value, (sequence_length, batch_size), "value")
    if sequence_operation not in _REDUCERS:
        raise ValueError(f"Invalid sequence operation {sequence_operation}")
    if batch_operation not in _REDUCERS:
        raise ValueError(f"Invalid batch operation {batch_operation}")
    per_sequence = _REDUCERS[sequence_operation](value, dim=0)
    return _REDUCERS[batch_operation](per_sequence, dim=0)


def aggregate_weights(sequence_length, batch_size, sequence_operation, batch_operation):
    """The constant c with aggregate(v) == c * v.sum(); lets a fused kernel
    fold the reduction into one scale factor."""
    if sequence_operation not in _REDUCERS:
        raise ValueError(f"Invalid sequence operation {sequence_operation}")
    if batch_operation not in _REDUCERS:
        raise ValueError(f"Invalid batch operation {batch_operation}")
    c = 1.0
    if sequence_operation == "mean":
        c /= sequence_length
    if batch_operation == "mean":
        c /= batch_size
    return c
