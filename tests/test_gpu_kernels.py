"""GPU parity tests proper: the HIP library (through its C ABI) against the
oracle, on the same seeded problems as the CPU math tests, plus larger batches
that span many wavefronts and a ragged last wavefront."""
import os

import pytest
import torch

import kernel_checks as KC
from kernel_harness import Backend

pytestmark = pytest.mark.gpu

DIMS = [(4, 2, 3), (2, 1, 2), (3, 2, 2), (8, 4, 4), (4, 2, 8)]


@pytest.fixture(scope="module", params=["lane", "quad"])
def be(request):
    """Every test runs against both small-state kernel families: one lane per
    sequence and four lanes per sequence (the library picks by batch size when
    the variable is unset; dimensions the quad form does not cover fall back)."""
    os.environ["KVAE_SEQ_LANES"] = "1" if request.param == "lane" else "4"
    yield Backend("hip")
    os.environ.pop("KVAE_SEQ_LANES", None)


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


@pytest.mark.parametrize("B", [1, 64, 130, 1000])
def test_filter_f32_batches(be, B):
    KC.check_filter(be, T=20, B=B, n=4, p=2, K=3, dtype=torch.float32, seed=9 + B,
                    mask_kind="binary", burn_in=3)


def test_filter_f64_many_waves(be):
    KC.check_filter(be, T=12, B=517, n=4, p=2, K=3, dtype=torch.float64, seed=21)


def test_filter_single_step(be):
    KC.check_filter(be, T=1, B=70, n=4, p=2, K=3, dtype=torch.float64, seed=3, mask_kind="binary",
                    per_sequence_carry=True)


@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("sample", [False, True])
def test_smooth_f64(be, n, p, K, sample):
    KC.check_smooth(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n, sample=sample)


@pytest.mark.parametrize("symmetrize,burn_in", [(False, 0), (True, 3), (False, 2)])
def test_smooth_options(be, symmetrize, burn_in):
    KC.check_smooth(be, T=7, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=11, sample=True,
                    symmetrize=symmetrize, burn_in=burn_in)


@pytest.mark.parametrize("B", [1, 130, 1000])
def test_smooth_f32_batches(be, B):
    KC.check_smooth(be, T=20, B=B, n=4, p=2, K=3, dtype=torch.float32, seed=2 + B, sample=True,
                    burn_in=3)


def test_smooth_T1(be):
    KC.check_smooth(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=2, sample=True)


@pytest.mark.parametrize("n,p,K", DIMS)
def test_elbo_f64(be, n, p, K):
    KC.check_elbo(be, T=6, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + 1)


@pytest.mark.parametrize("B", [1, 130, 1000])
def test_elbo_f32_batches(be, B):
    KC.check_elbo(be, T=20, B=B, n=4, p=2, K=3, dtype=torch.float32, seed=4 + B)


def test_elbo_T1(be):
    KC.check_elbo(be, T=1, B=2, n=4, p=2, K=3, dtype=torch.float64, seed=4)


# -- wavefront-per-sequence kernels (dimensions outside the lane list) ---------
WAVE_DIMS_F64 = [(6, 3, 2), (16, 8, 4)]


@pytest.mark.parametrize("n,p,K", WAVE_DIMS_F64)
@pytest.mark.parametrize("mask_kind", ["none", "fractional"])
def test_wave_filter_f64(be, n, p, K, mask_kind):
    KC.check_filter(be, T=5, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + p,
                    mask_kind=mask_kind, burn_in=1)


def test_wave_filter_options(be):
    KC.check_filter(be, T=6, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=5, mask_kind="binary",
                    symmetrize=False, burn_in=2, per_sequence_carry=True)
    KC.check_filter(be, T=1, B=2, n=16, p=8, K=4, dtype=torch.float64, seed=6, mask_kind="binary",
                    per_sequence_carry=True)


@pytest.mark.parametrize("n,p,K", WAVE_DIMS_F64)
@pytest.mark.parametrize("sample", [False, True])
def test_wave_smooth_f64(be, n, p, K, sample):
    KC.check_smooth(be, T=5, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n, sample=sample,
                    burn_in=1)


@pytest.mark.parametrize("n,p,K", WAVE_DIMS_F64)
def test_wave_elbo_f64(be, n, p, K):
    KC.check_elbo(be, T=5, B=3, n=n, p=p, K=K, dtype=torch.float64, seed=n + 1)


@pytest.fixture(params=["tile", "lds"])
def large_state(request):
    """Both large-state fp32 kernel families: matrices in registers (products on
    the 32x32 matrix-core tile) and matrices in LDS."""
    os.environ["KVAE_LARGE_STATE"] = request.param
    yield request.param
    os.environ.pop("KVAE_LARGE_STATE", None)


def test_tile_state16_f32(be, large_state):
    KC.check_filter(be, T=9, B=4, n=16, p=8, K=4, dtype=torch.float32, seed=41, mask_kind="binary",
                    burn_in=2)
    KC.check_smooth(be, T=9, B=4, n=16, p=8, K=4, dtype=torch.float32, seed=42, sample=True,
                    burn_in=2)
    KC.check_smooth(be, T=9, B=4, n=16, p=8, K=4, dtype=torch.float32, seed=43, sample=False,
                    symmetrize=False)
    KC.check_elbo(be, T=9, B=4, n=16, p=8, K=4, dtype=torch.float32, seed=44)
    KC.check_elbo(be, T=1, B=3, n=16, p=8, K=4, dtype=torch.float32, seed=45)
    # long enough for the kernels to cut a sequence into pieces (two of 33 and 32 steps)
    KC.check_elbo(be, T=65, B=2, n=16, p=8, K=4, dtype=torch.float32, seed=46)


def test_wave_large_state_f32(be, large_state):
    """BASELINE config 4 shapes (z_dim 32, a_dim 16, K 8) at a test-sized T."""
    KC.check_filter(be, T=12, B=5, n=32, p=16, K=8, dtype=torch.float32, seed=1, mask_kind="binary",
                    burn_in=2)
    KC.check_smooth(be, T=12, B=5, n=32, p=16, K=8, dtype=torch.float32, seed=2, sample=True,
                    burn_in=2)
    KC.check_elbo(be, T=12, B=5, n=32, p=16, K=8, dtype=torch.float32, seed=3)
    # three pieces of 34, 34 and 32 steps
    KC.check_elbo(be, T=100, B=3, n=32, p=16, K=8, dtype=torch.float32, seed=5)
    KC.check_elbo(be, T=1, B=2, n=6, p=3, K=2, dtype=torch.float64, seed=4)


def test_wave_long_sequence_drift_f32(be, large_state):
    """T = 1000 at z_dim 32 in float32 against the float64 oracle: rounding must
    not accumulate over the sequence (outputs and gradients stay within 1e-4)."""
    KC.check_filter(be, T=1000, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=31,
                    mask_kind="binary", burn_in=3)
    KC.check_smooth(be, T=1000, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=32, sample=True,
                    burn_in=3)
    KC.check_elbo(be, T=1000, B=2, n=32, p=16, K=8, dtype=torch.float32, seed=33)


def test_lane_long_sequence_drift_f32(be):
    KC.check_filter(be, T=1000, B=3, n=4, p=2, K=3, dtype=torch.float32, seed=34,
                    mask_kind="binary", burn_in=3)
    KC.check_smooth(be, T=1000, B=3, n=4, p=2, K=3, dtype=torch.float32, seed=35, sample=True)
    KC.check_elbo(be, T=1000, B=3, n=4, p=2, K=3, dtype=torch.float32, seed=36)


@pytest.mark.parametrize("n,p,K", [(4, 2, 3), (2, 1, 2), (3, 2, 2), (8, 4, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_masked_filter_with_network_in_kernel(n, p, K, dtype):
    """One-launch masked filter (two-layer perceptron inside the recursion) vs the oracle:
    binary and fractional masks, no mask, "mean" and "sample" imputation, ragged batch."""
    KC.check_masked_mlp("cuda:0", T=9, B=67, n=n, p=p, K=K, hidden=24, dtype=dtype, seed=n,
                        mask_kind="binary", sample=True)
    KC.check_masked_mlp("cuda:0", T=20, B=130, n=n, p=p, K=K, hidden=128, dtype=dtype, seed=n + 1,
                        mask_kind="fractional", sample=False, symmetrize=False)
    KC.check_masked_mlp("cuda:0", T=1, B=1, n=n, p=p, K=K, hidden=5, dtype=dtype, seed=n + 2,
                        mask_kind="none", sample=True)


def test_masked_filter_network_too_large_is_refused():
    from kalman_vae_amd import functions as F
    dev = torch.device("cuda:0")
    z = lambda *s: torch.zeros(*s, device=dev)
    hidden = 4096      # 4096 * 6 + 3 floats > 16 KB
    assert not F.masked_mlp_fits(hidden, 2, 3, torch.float32)
    out = F.masked_filter_mlp(z(2, 3, 2), z(2, 3), z(hidden * 6 + 3), hidden, z(3, 4, 4), z(3, 2, 4),
                              torch.eye(4, device=dev), torch.eye(2, device=dev), z(4),
                              torch.eye(4, device=dev))
    assert out is None
