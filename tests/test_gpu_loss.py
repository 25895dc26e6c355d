"""The fused objective kvae_kalman_loss_forward / _backward (filter -> smoother ->
ELBO terms -> weighted totals in one call per direction) on the HIP library, through the
C ABI, against the float64 oracle: per-pair terms, the four output scalars, d_a, d_alpha
and every parameter gradient.  float32 tolerance 1e-4 (relative, per tensor), float64
1e-9, as for the separate kernels."""
import os

import pytest
import torch

import kernel_checks as KC
from kernel_harness import Backend

pytestmark = pytest.mark.gpu

DIMS = [(4, 2, 3), (2, 1, 2), (3, 2, 2), (8, 4, 4), (4, 2, 8)]


@pytest.fixture(scope="module")
def be():
    return Backend("hip")


@pytest.fixture(params=["lane", "quad", "row"], autouse=True)
def family(request):
    """Run every case against all three small-state kernel families (one, four or sixteen
    lanes per sequence; the library picks by batch size unless KVAE_SEQ_LANES forces one;
    where the sixteen-lane form does not exist -- float64, n = 8, T > 36 -- "row" runs the
    library's own choice)."""
    os.environ["KVAE_SEQ_LANES"] = {"lane": "1", "quad": "4", "row": "16"}[request.param]
    yield request.param
    os.environ.pop("KVAE_SEQ_LANES", None)


@pytest.mark.parametrize("n,p,K", DIMS)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("mask_kind", ["none", "fractional"])
def test_loss_matches_oracle(be, n, p, K, dtype, mask_kind):
    KC.check_loss(be, T=8, B=37, n=n, p=p, K=K, dtype=dtype, seed=n + p, mask_kind=mask_kind,
                  burn_in=2)


@pytest.mark.parametrize("symmetrize,burn_in,stride", [(False, 0, 1), (True, 3, 0), (False, 2, 0),
                                                       (True, 30, 1)])
def test_loss_options(be, symmetrize, burn_in, stride):
    KC.check_loss(be, T=7, B=130, n=4, p=2, K=3, dtype=torch.float64, seed=13,
                  mask_kind="binary", symmetrize=symmetrize, burn_in=burn_in, stride=stride)
    KC.check_loss(be, T=7, B=130, n=4, p=2, K=3, dtype=torch.float32, seed=13,
                  mask_kind="binary", symmetrize=symmetrize, burn_in=burn_in, stride=stride)


@pytest.mark.parametrize("B", [1, 16, 63, 64, 65, 1000])
def test_loss_ragged_batches(be, B):
    KC.check_loss(be, T=20, B=B, n=4, p=2, K=3, dtype=torch.float32, seed=B, burn_in=3)


def test_loss_short_sequences(be):
    KC.check_loss(be, T=1, B=5, n=4, p=2, K=3, dtype=torch.float64, seed=6)
    KC.check_loss(be, T=2, B=5, n=4, p=2, K=3, dtype=torch.float64, seed=7, burn_in=1)
    # float32 as well: the 16-lane form exists for float only, and its peeled first and last
    # steps are all there is at T = 1, 2 and 3
    for T in (1, 2, 3):
        KC.check_loss(be, T=T, B=7, n=4, p=2, K=3, dtype=torch.float32, seed=40 + T,
                      mask_kind="binary" if T > 1 else "none")


@pytest.mark.parametrize("n,p,K", [(4, 2, 3), (2, 1, 2), (3, 2, 2), (4, 2, 8)])
def test_loss_row_form_full_lds(be, n, p, K):
    """T = 35 is the longest sequence whose per-step state (with what the gradient needs) fits
    the 16-lane form's 64 KB of LDS."""
    KC.check_loss(be, T=35, B=11, n=n, p=p, K=K, dtype=torch.float32, seed=31, burn_in=3,
                  mask_kind="binary")


def test_loss_long_sequence_f32(be):
    KC.check_loss(be, T=200, B=8, n=4, p=2, K=3, dtype=torch.float32, seed=21, burn_in=3)
