"""CPU check of the device math: the per-sequence source that the HIP kernels
execute (kalman_vae_amd/csrc/kalman_core.h) is compiled for the host by
tests/hostsim and compared, forward and adjoint, with the oracle.  This is how
the hand-derived adjoints are validated in a container without a GPU; the GPU
tests (test_gpu_kernels.py) repeat the same checks on the HIP library."""
import pytest
import torch

import kernel_checks as KC
from kernel_harness import Backend

DIMS = [(4, 2, 3), (2, 1, 2), (3, 2, 2), (8, 4, 4)]


@pytest.fixture(scope="module")
def be():
    return Backend("hostsim")


@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("mask_kind", ["none", "binary", "fractional"])
def test_filter_f64(be, n, p, K, mask_kind):
    KC.check_filter(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + p,
                    mask_kind=mask_kind)


@pytest.mark.parametrize("symmetrize,burn_in,carry", [(False, 0, False), (True, 2, False),
                                                      (True, 0, True), (False, 3, True)])
def test_filter_options(be, symmetrize, burn_in, carry):
    KC.check_filter(be, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=5, mask_kind="binary",
                    symmetrize=symmetrize, burn_in=burn_in, per_sequence_carry=carry)


def test_filter_f32(be):
    KC.check_filter(be, T=20, B=5, n=4, p=2, K=3, dtype=torch.float32, seed=9, mask_kind="binary")


def test_filter_single_step(be):
    KC.check_filter(be, T=1, B=4, n=4, p=2, K=3, dtype=torch.float64, seed=3, mask_kind="binary",
                    per_sequence_carry=True)


@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("sample", [False, True])
def test_smooth_f64(be, n, p, K, sample):
    KC.check_smooth(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n, sample=sample)


@pytest.mark.parametrize("symmetrize,burn_in", [(False, 0), (True, 3), (False, 2)])
def test_smooth_options(be, symmetrize, burn_in):
    KC.check_smooth(be, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=11, sample=True,
                    symmetrize=symmetrize, burn_in=burn_in)


def test_smooth_f32_and_T1(be):
    KC.check_smooth(be, T=20, B=4, n=4, p=2, K=3, dtype=torch.float32, seed=2, sample=True)
    KC.check_smooth(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=2, sample=True)


@pytest.mark.parametrize("n,p,K", DIMS)
def test_elbo_f64(be, n, p, K):
    KC.check_elbo(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + 1)


def test_elbo_f32_and_T1(be):
    KC.check_elbo(be, T=20, B=4, n=4, p=2, K=3, dtype=torch.float32, seed=4)
    KC.check_elbo(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=4)


# -- wavefront-per-sequence source (kalman_wave.h), same checks ----------------
WAVE_DIMS = [(4, 2, 3), (6, 3, 2), (16, 8, 4), (32, 16, 8)]


@pytest.fixture(scope="module")
def bw():
    return Backend("hostsim_wave")


@pytest.mark.parametrize("n,p,K", WAVE_DIMS)
@pytest.mark.parametrize("mask_kind", ["none", "fractional"])
def test_wave_filter_f64(bw, n, p, K, mask_kind):
    KC.check_filter(bw, T=5, B=2, n=n, p=p, K=K, dtype=torch.float64, seed=n + p,
                    mask_kind=mask_kind, burn_in=1)


def test_wave_filter_options(bw):
    KC.check_filter(bw, T=6, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=5, mask_kind="binary",
                    symmetrize=False, burn_in=2, per_sequence_carry=True)
    KC.check_filter(bw, T=1, B=2, n=16, p=8, K=4, dtype=torch.float64, seed=6, mask_kind="binary",
                    per_sequence_carry=True)


@pytest.mark.parametrize("n,p,K", WAVE_DIMS)
@pytest.mark.parametrize("sample", [False, True])
def test_wave_smooth_f64(bw, n, p, K, sample):
    KC.check_smooth(bw, T=5, B=2, n=n, p=p, K=K, dtype=torch.float64, seed=n, sample=sample,
                    burn_in=1)


def test_wave_smooth_options(bw):
    KC.check_smooth(bw, T=6, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=11, sample=True,
                    symmetrize=False, burn_in=2)
    KC.check_smooth(bw, T=1, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=12, sample=True)


@pytest.mark.parametrize("n,p,K", WAVE_DIMS)
def test_wave_elbo_f64(bw, n, p, K):
    KC.check_elbo(bw, T=5, B=2, n=n, p=p, K=K, dtype=torch.float64, seed=n + 1)


def test_wave_f32_large_state(bw):
    KC.check_filter(bw, T=8, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=1, mask_kind="binary")
    KC.check_smooth(bw, T=8, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=2, sample=True)
    KC.check_elbo(bw, T=8, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=3)
    KC.check_elbo(bw, T=1, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=4)
