"""The full model (frame encoder -> Kalman ELBO -> frame decoder) against
fixtures produced by the reference's ``KalmanVariationalAutoencoder.elbo(xs=...)``
(tests/golden/kvae/*.npz, made by tests/golden/make_golden_kvae.py).

The module must load the reference's state_dict unchanged, consume the recorded
standard-normal draws in the reference's order (encoder sample first, then
filter, smoother and the ELBO sample), and reproduce the objective, every
scalar term and the gradient of the objective w.r.t. every parameter.

Tolerances (max-norm relative error per tensor): float64 1e-7 (5e-7 against the
fixture at the frame size of examples/train.py, 64x64, which stores its float64
results in float32); float32 against the float64 fixture 1e-4 on the objective, its
terms and the Kalman-path gradients, 1e-3 on the gradients of the convolutional
encoder/decoder (stock fp32 MIOpen kernels, outside the accelerated path).

The 64x64 fixture leaves out the LSTM matrices (regenerated from its seed by
golden_util.seeded_tensor, as the generating script did) and keeps every 8th
element and the sum of their gradients.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "kvae", "*.npz")))
SCALARS = ["reconstruction", "regularization", "kl", "kalman_observation_log_likelihood",
           "kalman_state_transition_log_likelihood", "kalman_posterior_log_likelihood"]


def rel(got, want):
    got, want = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(want).double().cpu()
    return float(((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).detach())


def run(case, dtype, monkeypatch):
    import kalman_vae_amd.elbo as ke
    import kalman_vae_amd.kalman_vae as kv
    import kalman_vae_amd.state_space_model as ssm_mod
    from kalman_vae_amd import KalmanVariationalAutoencoder, SampleControl
    d = np.load(os.path.join(HERE, "golden", "kvae", case + ".npz"))
    cfg = json.loads(str(d["config"]))
    dev = torch.device("cuda:0")
    model = KalmanVariationalAutoencoder(
        image_size=tuple(cfg["image"]), image_channels=cfg["channels"], a_dim=cfg["p"],
        z_dim=cfg["n"], K=cfg["K"], init_transition_reg_weight=0.9,
        init_observation_reg_weight=0.9, learn_noise_covariance=True, init_noise_scale=0.7,
        dynamics_parameter_network=cfg["net"], decoder_type=cfg["decoder"])
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd.")}
    if cfg.get("seeded_lstm", False):
        from golden_util import seeded_tensor
        for k, v in model.state_dict().items():
            if ".lstm." in k:
                sd[k] = seeded_tensor(k, v.shape, cfg["seed"], 128 ** -0.5)
    model.load_state_dict(sd, strict=True)      # same names and shapes as the reference
    model = model.to(dtype).to(dev)
    model.train(cfg["train"])
    draws = [torch.from_numpy(d[f"eps.{i}"]).to(dtype).to(dev) for i in range(int(d["n_draws"]))]
    pos = [0]

    def replay(shape, dtype, device):
        e = draws[pos[0]]
        pos[0] += 1
        assert tuple(e.shape) == tuple(shape), (pos[0], tuple(e.shape), tuple(shape))
        return e

    for mod in (kv, ke, ssm_mod):
        monkeypatch.setattr(mod, "_standard_normal", replay)
    xs = torch.from_numpy(d["in.xs"]).to(dtype).to(dev)
    sc = SampleControl(encoder=cfg["encoder"], decoder="mean",
                       state_transition=cfg["state_transition"], observation=cfg["observation"])
    obj, info = model.elbo(
        sample_control=sc, xs=xs, kl_weight=cfg["kl_weight"], burn_in=cfg["burn_in"],
        reconst_weight=cfg.get("reconst_weight", 0.3),
        regularization_weight=cfg.get("regularization_weight", 1.0),
        kalman_weight=cfg.get("kalman_weight", 1.0),
        sequence_operation=cfg["sequence_operation"], batch_operation=cfg["batch_operation"])
    assert pos[0] == len(draws), "the module drew fewer random tensors than the reference"
    named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(obj, [p for _, p in named], allow_unused=True)
    return d, obj, info, {k: g for (k, _), g in zip(named, grads)}


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_full_model_matches_reference(case, dtype, monkeypatch):
    d, obj, info, grads = run(case, dtype, monkeypatch)
    cfg = json.loads(str(d["config"]))
    tol = (5e-7 if cfg.get("store_f32", False) else 1e-7) if dtype == torch.float64 else 1e-4
    stride = cfg.get("grad_stride", 1)
    bad = []

    def check(name, got, want, t=tol):
        e = rel(got, want)
        if not e < t:
            bad.append(f"{name}: {e:.2e}")

    check("objective", obj, d["out.objective"])
    for k in SCALARS:
        check(k, info[k], d["out." + k])
    check("as", info["as"], d["out.as"])
    check("means", info["means"], d["out.means"])
    for k, g in grads.items():
        if "gradsub." + k in d.files:
            assert g is not None, k
            check("grad (every %dth) %s" % (stride, k), g.reshape(-1)[::stride], d["gradsub." + k])
            scale = float(g.abs().sum())
            assert abs(float(g.sum()) - float(d["gradsum." + k])) < 10 * tol * scale, k
            continue
        want = d["grad." + k] if "grad." + k in d.files else None
        if want is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        stock = k.startswith(("encoder.", "decoder."))
        check("grad " + k, g, want, tol if dtype == torch.float64 or not stock else 1e-3)
    assert not bad, "; ".join(bad)


def test_argument_errors():
    from kalman_vae_amd import KalmanVariationalAutoencoder, SampleControl
    m = KalmanVariationalAutoencoder((16, 16), 1, 2, 4, 3, 0.9, 0.9, False, 1.0, "mlp").cuda()
    sc = SampleControl()
    with pytest.raises(ValueError):
        m.elbo(sc)
    with pytest.raises(ValueError):
        m.elbo(sc, xs=torch.zeros(2, 1, 1, 16, 16).cuda(), as_=torch.zeros(2, 1, 2).cuda())
    with pytest.raises(ValueError):
        KalmanVariationalAutoencoder((16, 16), 1, 2, 4, 3, 0.9, 0.9, False, 1.0, "mlp", "poisson")
    m.train()
    with pytest.raises(ValueError):     # the encoder mean is an inference-only choice
        m.elbo(SampleControl(encoder="mean"), xs=torch.zeros(2, 1, 1, 16, 16).cuda())


def test_evaluation_sweeps(tmp_path):
    """The imputation sweeps (kvae/evaluation.py:94-207) on a synthetic batch: table layout,
    value ranges, nan where nothing is masked, and the same numbers whether the masked
    filter ran its inference path (the sweeps, no_grad) or its autograd path."""
    import math
    from kalman_vae_amd import KalmanVariationalAutoencoder, SampleControl
    from kalman_vae_amd import evaluation as ev
    torch.manual_seed(2)
    dev = torch.device("cuda:0")
    model = KalmanVariationalAutoencoder((16, 16), 1, 2, 4, 3, 0.9, 0.9, False, 1.0, "lstm",
                                         decoder_type="bernoulli").to(dev).eval()
    frames = torch.rand(14, 6, 1, 16, 16)
    loader = [frames]
    sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")
    rnd, cont = ev.evaluate(loader, model, sc, str(tmp_path), 3, torch.float32, dev)
    assert list(rnd.columns) == ["batch_id", "dropout_probabilities", "filtering_incorrect_pixels",
                                 "smoothing_incorrect_pixels"]
    assert list(cont.columns) == ["batch_id", "mask_lengths", "filtering_incorrect_pixels",
                                  "smoothing_incorrect_pixels"]
    assert len(rnd) == 11 and cont["mask_lengths"].tolist() == [2, 4, 6, 8]
    assert math.isnan(rnd["filtering_incorrect_pixels"][0])          # dropout 0: nothing masked
    for col in ("filtering_incorrect_pixels", "smoothing_incorrect_pixels"):
        assert all(0.0 <= v <= 1.0 for v in cont[col])
        assert all(0.0 <= v <= 1.0 for v in rnd[col][1:])
    assert (tmp_path / "tables" / "epoch_3" / "random_masking.csv").exists()
    assert (tmp_path / "tables" / "epoch_3" / "continuous_masking.csv").exists()

    batch = (frames > 0.5).float().to(dev)
    mask = ev.create_continuous_mask(14, 6, 6, dev, torch.float32)
    fast = ev._incorrect_pixels(model, batch, mask, sc)
    _, info = model.elbo(xs=batch, observation_mask=mask, sample_control=sc)      # autograd path
    slow = []
    for key in ("filter_as", "as_resampled"):
        rec = model.decoder(info[key].reshape(14 * 6, -1)).mean.view(batch.shape)
        slow.append(float(ev.calculate_fraction_of_incorrect_pixels(batch, rec, mask)))
    assert all(abs(a - b) < 1e-6 for a, b in zip(fast, slow)), (fast, slow)
