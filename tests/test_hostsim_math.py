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
