"""Imputation benchmarks of the reference's evaluation module (kvae/evaluation.py:21-207,
272-286): masks, the fraction of wrongly reconstructed pixels at the masked steps, and
the two sweeps (random dropout, a contiguous gap) over one batch of a dataloader.

The sweeps run under ``torch.no_grad()``, which lets the masked filter take its
inference path (one HIP launch per step for the imputation, the persistent LSTM
kernel continuing from its carried state).  Video rendering and wandb logging of
the reference (kvae/evaluation.py:210-436) are not reproduced.

Parity note: the reference module imports wandb and moviepy, which this build
environment lacks, so it cannot be executed here to produce fixtures; the two mask
builders and the pixel metric are checked against hand-computed values and the
sweeps against their definition (tests/test_cpu_host_logic.py,
tests/test_gpu_kvae_model.py).
"""
import os

import pandas as pd
import torch

from .sample_control import SampleControl

DROPOUT_PROBABILITIES = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1]


def create_continuous_mask(seq_length, mask_length, batch_size, device, dtype):
    """(T,B) ones with a centred gap of ``mask_length`` zeros (evaluation.py:21-31)."""
    start = (seq_length - mask_length) // 2
    column = torch.ones(seq_length)
    column[start:start + mask_length] = 0.0
    return column.unsqueeze(1).repeat(1, batch_size).to(device=device, dtype=dtype)


def create_random_mask(seq_length, batch_size, mask_rate, device, dtype):
    """(T,B) Bernoulli keep mask with the first and last step always observed
    (evaluation.py:34-41; one ``torch.rand`` draw on ``device``)."""
    mask = (torch.rand((seq_length, batch_size), device=device) >= mask_rate).to(dtype)
    mask[0] = 1
    mask[-1] = 1
    return mask


def calculate_fraction_of_incorrect_pixels(image, reconstructed_image, observation_mask):
    """Share of pixels of the masked steps where the thresholded reconstruction differs
    from the binary image (evaluation.py:272-286; the reference hard-codes 16x16 frames,
    here the mask is broadcast to whatever the frame shape is).  0/0 = nan when nothing
    is masked, as in the reference."""
    wrong = (image != (reconstructed_image > 0.5)).to(image.dtype)
    hidden = 1.0 - observation_mask.to(dtype=image.dtype, device=image.device)
    hidden = hidden.view(*hidden.shape, *([1] * (image.dim() - hidden.dim()))).expand_as(image)
    return (wrong * hidden).sum() / hidden.sum()


def _first_batch(dataloader, dtype, device):
    return (next(iter(dataloader)) > 0.5).to(dtype=dtype, device=device)


@torch.no_grad()
def _incorrect_pixels(kvae, batch, mask, sample_control):
    # the sweeps read two tensors of the reference's elbo() call (kvae/evaluation.py:100-140):
    # the filter's imputed observations and the smoother's re-generated ones
    T, B = batch.shape[:2]
    _, as_ = kvae._encode(batch, sample_control)
    imputed = kvae.state_space_model.impute(as_, sample_control, observation_mask=mask)
    result = []
    for a in imputed:
        frames = kvae.decoder(a.reshape(T * B, -1)).mean.view(batch.shape)
        result.append(float(calculate_fraction_of_incorrect_pixels(batch, frames, mask)))
    return result


def evaluate_random_masking(dataloader, kvae, sample_control: SampleControl, dtype, device):
    """evaluation.py:94-147: a fresh batch and a fresh random mask per dropout probability."""
    filtering, smoothing = [], []
    for rate in DROPOUT_PROBABILITIES:
        batch = _first_batch(dataloader, dtype, device)
        mask = create_random_mask(batch.shape[0], batch.shape[1], rate, batch.device, batch.dtype)
        f, s = _incorrect_pixels(kvae, batch, mask, sample_control)
        filtering.append(f)
        smoothing.append(s)
    return pd.DataFrame({"batch_id": [0] * len(DROPOUT_PROBABILITIES),
                         "dropout_probabilities": DROPOUT_PROBABILITIES,
                         "filtering_incorrect_pixels": filtering,
                         "smoothing_incorrect_pixels": smoothing})


def evaluate_continuous_masking(dataloader, kvae, sample_control: SampleControl, dtype, device):
    """evaluation.py:150-207: one batch, centred gaps of 2, 4, ... < T - 4 steps."""
    batch = _first_batch(dataloader, dtype, device)
    T, B = batch.shape[:2]
    mask_lengths = list(range(2, T - 4, 2))
    filtering, smoothing = [], []
    for length in mask_lengths:
        mask = create_continuous_mask(T, length, B, batch.device, batch.dtype)
        f, s = _incorrect_pixels(kvae, batch, mask, sample_control)
        filtering.append(f)
        smoothing.append(s)
    return pd.DataFrame({"batch_id": 0, "mask_lengths": mask_lengths,
                         "filtering_incorrect_pixels": filtering,
                         "smoothing_incorrect_pixels": smoothing})


def evaluate(dataloader, kvae, sample_control, checkpoint_dir, epoch, dtype, device):
    """Both sweeps, written as the reference writes them
    (``<checkpoint_dir>/tables/epoch_<epoch>/{random,continuous}_masking.csv``,
    evaluation.py:44-91 without the videos).  Returns the two tables."""
    random_masking = evaluate_random_masking(dataloader, kvae, sample_control, dtype, device)
    continuous_masking = evaluate_continuous_masking(dataloader, kvae, sample_control, dtype, device)
    table_directory = os.path.join(checkpoint_dir, "tables", f"epoch_{epoch}")
    os.makedirs(table_directory, exist_ok=True)
    random_masking.to_csv(os.path.join(table_directory, "random_masking.csv"))
    continuous_masking.to_csv(os.path.join(table_directory, "continuous_masking.csv"))
    return random_masking, continuous_masking
