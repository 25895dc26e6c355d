"""CPU-side tests: the C-ABI library loads and exports every declared symbol,
the ctypes mirrors match the C structs, and the host logic around the kernels
(weight networks, parameter properties, state_dict layout, aggregation,
error behaviour) matches the reference as recorded in the golden fixtures.
No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest
import torch

from golden_util import Golden, golden_names, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kvae_kalman.h")


def test_library_exports_every_declared_symbol():
    from kalman_vae_amd import _capi as capi
    declared = set(re.findall(r"\b(kvae_[a-z0-9_]+)\s*\(", open(HEADER).read()))
    assert declared == set(capi.EXPORTS)
    lib = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_host_only_entry_points():
    from kalman_vae_amd import _capi as capi
    lib = capi.load()
    assert lib.kvae_version() == 1
    assert lib.kvae_supported(4, 2, 3, capi.KVAE_F32) == 1
    assert lib.kvae_supported(4, 2, 3, capi.KVAE_F64) == 1
    assert lib.kvae_supported(5, 2, 3, capi.KVAE_F32) == 0
    assert lib.kvae_supported(4, 2, 3, 7) == 0
    d = capi.Dims(20, 4096, 4, 2, 3, capi.KVAE_F32, 1, 0)
    rows = 4096 // 16   # one row of partial sums per wavefront; the quad form holds 16 sequences
    assert lib.kvae_workspace_bytes(C.byref(d), capi.OP_FILTER_BWD) == rows * (48 + 24 + 16 + 4) * 4
    assert lib.kvae_workspace_bytes(C.byref(d), capi.OP_SMOOTH_BWD) == rows * 4 * 4
    assert lib.kvae_workspace_bytes(C.byref(d), 99) == 0
    assert b"workspace" in lib.kvae_error_string(-3)
    # slabs of the LSTM weight gradient: at most 64, each a whole number of 64-row steps,
    # together covering the rows; a small batch gets one-step slabs
    assert lib.kvae_lstm128_wgrad_slabs(0) == 0
    assert lib.kvae_lstm128_wgrad_slabs(640) == 10 and lib.kvae_lstm128_wgrad_slabs(81920) == 64
    for R in (1, 63, 64, 65, 111, 640, 4097, 7000, 81920, 1024000, 10 ** 9):
        S = lib.kvae_lstm128_wgrad_slabs(R)
        rps = -(-(-(-R // S)) // 64) * 64          # rows per slab as the launcher rounds them
        assert 1 <= S <= 64 and S * rps >= R and (S - 1) * rps < R, (R, S, rps)
    # argument validation happens before any launch
    io = capi.FilterFwd()
    assert lib.kvae_filter_forward(C.byref(d), C.byref(io), None) == -1
    bad = capi.Dims(0, 4, 4, 2, 3, 0, 1, 0)
    assert lib.kvae_filter_forward(C.byref(bad), C.byref(io), None) == -1


def test_ctypes_structs_match_header():
    from kalman_vae_amd import _capi as capi
    names = {"kvae_dims": capi.Dims, "kvae_filter_fwd": capi.FilterFwd,
             "kvae_filter_bwd": capi.FilterBwd, "kvae_smooth_fwd": capi.SmoothFwd,
             "kvae_smooth_bwd": capi.SmoothBwd, "kvae_elbo_fwd": capi.ElboFwd,
             "kvae_elbo_bwd": capi.ElboBwd}
    src = '#include <stdio.h>\n#include "kvae_kalman.h"\nint main(void){\n'
    for n in names:
        src += f'printf("{n} %zu\\n", sizeof({n}));\n'
    src += 'printf("off_status %zu\\n", offsetof(kvae_filter_fwd, status));\n'
    src += 'printf("off_dLR %zu\\n", offsetof(kvae_smooth_bwd, d_LR));\n'
    src += 'printf("off_dP0 %zu\\n", offsetof(kvae_elbo_bwd, d_P0));\nreturn 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        cfile = os.path.join(td, "s.c")
        open(cfile, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])
        out = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for n, cls in names.items():
        assert int(out[n]) == C.sizeof(cls), n
    assert int(out["off_status"]) == capi.FilterFwd.status.offset
    assert int(out["off_dLR"]) == capi.SmoothBwd.d_LR.offset
    assert int(out["off_dP0"]) == capi.ElboBwd.d_P0.offset


def test_no_cpu_path():
    from kalman_vae_amd import SampleControl, StateSpaceModel
    ssm = StateSpaceModel(2, 4, 3, "mlp").eval()
    a = torch.randn(5, 3, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ssm.kalman_filter(a, SampleControl(state_transition="mean", observation="mean"))


@pytest.mark.parametrize("name", golden_names("g*"))
def test_state_dict_and_weight_network_match_reference(name):
    """Reference checkpoints load key for key, and the weight network fed with
    the reference's network inputs reproduces the reference's mixing weights."""
    from kalman_vae_amd import StateSpaceModel
    g = Golden(name)
    c = g.cfg
    ssm = StateSpaceModel(a_dim=c["p"], z_dim=c["n"], K=c["K"],
                          dynamics_parameter_network=c["net"],
                          hidden_dim=c.get("hidden_dim", 8), num_layers=c.get("num_layers", 2),
                          Q_reg=float(g["in.Q_reg"]), R_reg=float(g["in.R_reg"])).to(g.dtype)
    sd = {k[3:]: v for k, v in g.t.items() if k.startswith("sd.")}
    assert set(sd) == set(ssm.state_dict())
    ssm.load_state_dict(sd)
    tol = 1e-12 if g.dtype == torch.float64 else 1e-5
    if g.stored_f32:
        tol = 1e-6   # float64 run stored as float32
    assert rel_err(ssm.mat_Q, g["in.Q"]) < tol
    assert rel_err(ssm.mat_R, g["in.R"]) < tol
    # one call over the whole sequence == the reference's step-by-step calls
    ssm.weight_model.clear_hidden_state()
    w = ssm.weight_model(g["out.filter_as"])
    got = torch.cat([torch.full_like(w[:1], 1.0 / c["K"]), w[:-1]])
    assert rel_err(got, g["out.weights"]) < (1e-10 if g.dtype == torch.float64
                                             and not g.stored_f32 else 1e-5)


def test_matrix_properties_gates_and_setters():
    from kalman_vae_amd import StateSpaceModel
    ssm = StateSpaceModel(2, 4, 3, "mlp")
    assert not ssm.mat_Q.requires_grad and not ssm.mat_R.requires_grad
    assert ssm.mat_A_K.requires_grad and ssm.mat_C_K.requires_grad
    assert torch.allclose(ssm.mat_Q, 1.001 * torch.eye(4))
    ssm.learn_noise_covariance = True
    assert ssm.mat_Q.requires_grad
    ssm.fix_matrices = True
    assert not ssm.mat_Q.requires_grad and not ssm.mat_A_K.requires_grad
    ssm.mat_Q = 2.0 * torch.eye(4)
    assert ssm.Q_reg == 0.0 and torch.allclose(ssm.mat_Q, 2.0 * torch.eye(4))
    with pytest.raises(ValueError):
        ssm.mat_R = torch.eye(3)
    with pytest.raises(ValueError):
        StateSpaceModel(2, 4, 3, "mlp", initial_state_mean=torch.zeros(3))
    # eye/initial-state tensors follow .to() but stay out of the state_dict
    ssm = ssm.double()
    assert ssm._z_eye.dtype == torch.float64
    assert ssm.initial_state_covariance.dtype == torch.float64
    assert "initial_state_mean" not in ssm.state_dict()


def test_aggregate_and_sample_control():
    from kalman_vae_amd import SampleControl, aggregate
    from kalman_vae_amd.misc import aggregate_weights
    v = torch.arange(12.0).reshape(3, 4)
    for so in ("mean", "sum"):
        for bo in ("mean", "sum"):
            want = getattr(getattr(v, so)(0), bo)(0)
            assert torch.allclose(aggregate(v, 3, 4, so, bo), want)
            assert abs(aggregate_weights(3, 4, so, bo) * float(v.sum()) - float(want)) < 1e-9
    with pytest.raises(ValueError):
        aggregate(v, 4, 3, "mean", "mean")
    with pytest.raises(ValueError):
        aggregate(v, 3, 4, "max", "mean")
    sc = SampleControl()
    assert (sc.encoder, sc.decoder, sc.state_transition, sc.observation) == (
        "sample", "mean", "sample", "sample")


@pytest.mark.parametrize("case", ["gaussian_lstm_f64", "bernoulli_mlp_f64"])
def test_full_model_loads_reference_state_dict(case):
    """KalmanVariationalAutoencoder has the reference's parameter names and shapes
    (fixtures made from the reference's own module), and rejects bad arguments
    before touching the GPU library."""
    import json
    import numpy as np
    from kalman_vae_amd import KalmanVariationalAutoencoder, SampleControl
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "kvae", case + ".npz"))
    cfg = json.loads(str(d["config"]))
    model = KalmanVariationalAutoencoder(
        image_size=tuple(cfg["image"]), image_channels=cfg["channels"], a_dim=cfg["p"],
        z_dim=cfg["n"], K=cfg["K"], init_transition_reg_weight=0.9,
        init_observation_reg_weight=0.9, learn_noise_covariance=True, init_noise_scale=0.7,
        dynamics_parameter_network=cfg["net"], decoder_type=cfg["decoder"])
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd.")}
    assert set(sd) == set(model.state_dict())
    model.load_state_dict(sd, strict=True)
    with pytest.raises(ValueError):
        model.elbo(SampleControl())
    # encoder output on CPU is plain PyTorch: compare with the reference's encoded mean
    if cfg["encoder"] == "mean":
        model = model.double().eval()
        xs = torch.from_numpy(d["in.xs"])
        q = model.encoder(xs.reshape(-1, *xs.shape[2:]))
        assert rel_err(q.loc.view(cfg["T"], cfg["B"], -1), torch.from_numpy(d["out.as"])) < 1e-12


def test_checkpoint_layout_round_trip(tmp_path):
    """Checkpoints use the reference's dictionary layout (examples/train.py:153-164): a
    file written here restores model, optimizer and scheduler, and a file with the
    reference's keys built from a reference-generated state_dict loads strictly."""
    import json
    import numpy as np
    from kalman_vae_amd import KalmanVariationalAutoencoder
    from kalman_vae_amd.checkpoint import KEYS, load_checkpoint, save_checkpoint

    def make():
        torch.manual_seed(0)
        m = KalmanVariationalAutoencoder((16, 16), 1, 2, 4, 3, 0.9, 0.9, True, 0.7, "lstm")
        opt = torch.optim.Adam(m.parameters(), lr=3e-3)
        return m, opt, torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)

    m1, o1, s1 = make()
    for p in m1.parameters():
        p.grad = torch.ones_like(p)
    o1.step()
    s1.step()
    path = save_checkpoint(str(tmp_path), 7, m1, o1, s1, train_loss=1.5, test_loss=2.5)
    assert os.path.basename(path) == "state-7.pth"
    assert set(torch.load(path, weights_only=True)) == set(KEYS)
    m2, o2, s2 = make()
    meta = load_checkpoint(path, m2, o2, s2)
    assert meta == {"epoch": 7, "train_loss": 1.5, "test_loss": 2.5}
    for a, b in zip(m1.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    assert s2.last_epoch == s1.last_epoch and o2.state_dict()["state"].keys() == o1.state_dict()["state"].keys()
    # read the file the way the reference's evaluate.py:60-64 does: all three unconditionally
    m4, o4, s4 = make()
    ck = torch.load(path, weights_only=True)
    m4.load_state_dict(ck["model_state_dict"])
    o4.load_state_dict(ck["optimizer_state_dict"])
    s4.load_state_dict(ck["scheduler_state_dict"])
    assert s4.last_epoch == s1.last_epoch
    with pytest.warns(UserWarning, match="optimizer"):
        save_checkpoint(str(tmp_path / "bare"), 1, m1)

    # a file in the reference's layout whose state_dict came from the reference's module
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "kvae", "gaussian_lstm_f64.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd.")}
    ref_file = str(tmp_path / "state-0.pth")
    torch.save({"epoch": 0, "model_state_dict": sd, "optimizer_state_dict": None,
                "scheduler_state_dict": None, "train_loss": 0.0, "test_loss": 0.0}, ref_file)
    cfg = json.loads(str(d["config"]))
    m3 = KalmanVariationalAutoencoder(tuple(cfg["image"]), cfg["channels"], cfg["p"], cfg["n"],
                                      cfg["K"], 0.9, 0.9, True, 0.7, cfg["net"], cfg["decoder"]).double()
    load_checkpoint(ref_file, m3)
    assert torch.equal(m3.encoder.conv1.weight, sd["encoder.conv1.weight"])


def test_evaluation_masks_and_pixel_metric():
    """Mask builders and the incorrect-pixel fraction of the evaluation sweeps against
    hand-computed values (kvae/evaluation.py:21-41, 272-286)."""
    from kalman_vae_amd.evaluation import (calculate_fraction_of_incorrect_pixels,
                                           create_continuous_mask, create_random_mask)
    m = create_continuous_mask(10, 4, 3, "cpu", torch.float64)
    assert m.shape == (10, 3) and m.dtype == torch.float64
    assert m[:, 1].tolist() == [1, 1, 1, 0, 0, 0, 0, 1, 1, 1]      # gap starts at (10 - 4) // 2
    assert create_continuous_mask(7, 2, 1, "cpu", torch.float32)[:, 0].tolist() == [1, 1, 0, 0, 1, 1, 1]
    torch.manual_seed(5)
    r = create_random_mask(6, 50, 0.5, "cpu", torch.float32)
    torch.manual_seed(5)
    want = (torch.rand(6, 50) >= 0.5).float()
    want[0] = 1
    want[-1] = 1
    assert torch.equal(r, want) and 0.2 < float(r[1:-1].mean()) < 0.8
    assert float(create_random_mask(4, 3, 1.0, "cpu", torch.float32)[1:-1].sum()) == 0.0
    # T=2, B=1, one 1x2x2 frame per step; only step 1 is masked: 1 of its 4 pixels is wrong
    image = torch.tensor([[[[[1., 0.], [0., 1.]]]], [[[[1., 1.], [0., 0.]]]]])
    recon = torch.tensor([[[[[.1, .9], [.9, .1]]]], [[[[.9, .8], [.7, .2]]]]])
    mask = torch.tensor([[1.], [0.]])
    assert float(calculate_fraction_of_incorrect_pixels(image, recon, mask)) == 0.25
    assert torch.isnan(calculate_fraction_of_incorrect_pixels(image, recon, torch.ones(2, 1)))


def test_stepping_network_applies_only_on_the_gpu():
    """The per-step form of the LSTM network (three launches a step, used by the masked
    filter without a gradient) exists only for float32 tensors on a GPU: on the CPU the
    module offers no stepper, so the masked filter keeps calling it once per step, and a
    step-by-step call still equals the whole-sequence call (the carried state)."""
    from kalman_vae_amd.dynamics_parameter_network import LSTMModel
    torch.manual_seed(0)
    net = LSTMModel(2, 3, hidden_dim=128, num_layers=2).eval()
    x = torch.randn(5, 4, 2)
    with torch.no_grad():
        assert net.inference_stepper(x[0]) is None
        net.clear_hidden_state()
        whole = net(x)
        net.clear_hidden_state()
        steps = torch.cat([net(x[t:t + 1]) for t in range(5)])
    assert torch.allclose(whole, steps, atol=1e-6)
    assert abs(float(whole.sum()) - 5 * 4) < 1e-4        # rows of softmax weights


def test_undone_draws_are_served_again_in_order():
    """Noise drawn by a path that then does not run is queued for the path that runs instead:
    same tensors, same order, the generator consumed once."""
    from kalman_vae_amd import StateSpaceModel
    ssm = StateSpaceModel(2, 4, 3, "mlp", hidden_dim=8)
    like = torch.zeros(1)
    torch.manual_seed(3)
    with ssm._recording_draws() as log:
        first = [ssm._draw((5, 4), like), ssm._draw((5, 2), like), ssm._draw((5, 4), like)]
    assert len(log) == 3 and ssm._draw_log is None
    ssm._undo_draws(log)
    again = [ssm._draw((5, 4), like), ssm._draw((5, 2), like), ssm._draw((5, 4), like)]
    assert all(x is y for x, y in zip(first, again)) and not ssm._noise_pool
    fresh = ssm._draw((5, 4), like)
    torch.manual_seed(3)
    want = [torch.randn(5, 4), torch.randn(5, 2), torch.randn(5, 4), torch.randn(5, 4)]
    for x, y in zip(first + [fresh], want):
        assert torch.equal(x, y)
    # a queue that was installed before (all the noise of a step, drawn at once) stays behind
    ssm._noise_pool = [torch.ones(2, 2)]
    with ssm._recording_draws() as log:
        ssm._draw((2, 2), like)
    ssm._noise_pool = [torch.zeros(1, 1)]
    try:                                     # recording stops when the block fails, too
        with ssm._recording_draws():
            raise KeyError("x")
    except KeyError:
        pass
    assert ssm._draw_log is None
    ssm._undo_draws(log)
    assert [tuple(t.shape) for t in ssm._noise_pool] == [(2, 2), (1, 1)]
