#!/usr/bin/env python3
"""Benchmark of the Kalman filter + RTS smoother + Kalman-ELBO hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4|c5]

--config selects one of the five benchmark configurations (default c2, the one the
headline metric is quoted on):
  c1  training step, T=20, a_dim=2, z_dim=4, K=3, 32 sequences (the CPU reference's size)
  c2  the same shapes, 4096 sequences per GPU
  c3  masked imputation (filter + smoother, forward only, observation mask with keep
      probability 0.7), 1 048 576 sequences split over the GPUs, no collective
  c4  long / large state: T=1000, a_dim=16, z_dim=32, K=8, 1024 sequences per GPU
  c5  full Kalman-VAE training step on 64x64 frames, 128 sequences per GPU, Adam

One "step" of c1/c2/c4 = one forward+backward pass of the path over one batch of
synthetic bouncing-ball-shaped sequences: weight network -> mixing weights -> filter ->
smoother -> ELBO terms -> aggregate -> backward through all of it (plus, for N > 1, one
flat-bucket RCCL all-reduce of the parameter gradients).  Inputs are resident in HBM before
the timed region.  Per-GPU work is fixed as N grows (weak scaling), except c3 whose total
is fixed (strong scaling).

Rank 0 prints ONE JSON line.  `value` = sequence-steps/s of the whole job.  Additional
objects:
  roofline         the kernel that takes most of the step: achieved rate against its roof
  kalman_roofline  the longest kernel of the Kalman path itself, same form
  cpu_baseline     the oracle (a per-step PyTorch restatement of the reference algorithm)
                   timed on this machine's host cores
  kernels          per-kernel average durations (HIP events) and rates
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
VALU_FP32_PEAK_TF = 157.3   # MI355X vector fp32 spec peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 matrix-instruction peak


def bouncing_ball_observations(T, B, p, device, seed):
    """a_t = x_t + N(0, 0.05^2) for p/2 independent balls bouncing elastically in
    [-1,1]^2: x_0 ~ U(-0.8,0.8)^2, heading uniform, speed U(0.05,0.15) per step."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    balls = max(1, p // 2)
    x = (torch.rand(B, balls, 2, generator=g) * 1.6 - 0.8)
    ang = torch.rand(B, balls, generator=g) * 2 * math.pi
    speed = 0.05 + 0.10 * torch.rand(B, balls, generator=g)
    v = torch.stack([ang.cos(), ang.sin()], -1) * speed.unsqueeze(-1)
    out = []
    for _ in range(T):
        x = x + v
        over, under = x > 1.0, x < -1.0
        x = torch.where(over, 2.0 - x, x)
        x = torch.where(under, -2.0 - x, x)
        v = torch.where(over | under, -v, v)
        out.append(x.reshape(B, -1)[:, :p] + 0.05 * torch.randn(B, p, generator=g))
    return torch.stack(out).to(device)


def build_model(device, dtype, net, n=4, p=2, K=3, fused=True):
    from kalman_vae_amd import StateSpaceModel
    ssm = StateSpaceModel(a_dim=p, z_dim=n, K=K, dynamics_parameter_network=net)
    ssm.mat_Q = torch.eye(n)   # examples/train.py:73-74 of the reference
    ssm.mat_R = torch.eye(p)
    ssm = ssm.to(dtype).to(device).train()
    ssm.rng_parity = False      # one noise tensor per call instead of two per step
    ssm.check_numerics = False  # no host sync inside the step
    ssm.fused_loss = fused      # filter -> smoother -> ELBO terms: one launch per direction
    return ssm


# algorithmic work per sequence-step (n=z_dim, p=a_dim, K bases), forward
def flops_forward(n, p, K):
    filt = 2 * K * (n * n + p * n) + 4 * n**3 + 4 * p * n * n + 4 * p * p * n + 2 * p**3 \
        + 6 * p * n + 14 * n * n + 2 * p * p + 4 * n + 4 * p
    smooth = 10.67 * n**3 + 7 * n * n + 2 * p * n + 2 * n
    elbo = 5 * n * n + 2 * p * n + 2 * p * p + 6 * n
    return {"filter": filt, "smooth": smooth, "elbo": elbo}


def kernel_bytes(n, p, K, el):
    """Compulsory global-memory bytes per sequence-step of each kernel: every
    input element read once, every output element written once."""
    m, c, A, Cm = n, n * n, n * n, p * n
    return {
        "filter_forward": el * ((p + K) + (2 * m + 2 * c + A + Cm)),
        # reads a, alpha, saved m/P/m_pred/P_pred, upstream grads of the four mean/cov
        # outputs and of mat_As/mat_Cs; writes d_a, d_alpha
        "filter_backward": el * ((p + K + 2 * m + 2 * c) + (2 * m + 2 * c + A + Cm) + (p + K)),
        "smooth_forward": el * ((2 * m + 2 * c + A + Cm + n + p) + (m + c + n + p)),
        "smooth_backward": el * ((2 * m + 2 * c + A + Cm + n + p + m + c) + (m + c + n + p)
                                 + (2 * m + 2 * c + A + Cm)),
        "elbo_forward": el * ((p + m + c + A + Cm + n) + 3 + n),
        "elbo_backward": el * ((p + m + c + A + Cm + n) + 3 + (p + m + c + A + Cm)),
    }


MFMA_FP32_PEAK_TF = 157.3   # dense fp32 matrix-core peak of MI355X


def _timed(fn, iters):
    """Average seconds per call of fn (HIP events on the stream fn launches on)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def time_lstm_kernels(ssm, a, iters=20):
    """HIP-event durations of the weight network's recurrent kernels (one launch per layer
    and direction, recurrent matrix resident in registers) at the bench shape, launched
    through the C ABI on the current stream; None when the network is not the default
    two-layer H=128 LSTM."""
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd import lstm_function as LF
    net = ssm.weight_model
    lstm = getattr(net, "lstm", None)
    if lstm is None or not LF.persistent_supported(a, lstm):
        return None
    lib = capi.load()
    T, B, P = a.shape
    H = 128
    dev = a.device
    x = a.detach().contiguous()
    w = lambda nme: getattr(lstm, nme).detach().contiguous()
    new = lambda *s: torch.empty(*s, dtype=x.dtype, device=dev)
    gates, h, c = new(T, B, 4 * H), new(T, B, H), new(T, B, H)
    gx = torch.randn(T, B, 4 * H, dtype=x.dtype, device=dev)
    dG, dx = new(T, B, 4 * H), new(T, B, P)
    gh = torch.randn(T, B, H, dtype=x.dtype, device=dev) * 1e-3
    part = new(lib.kvae_lstm128_blocks(B), 4 * H * (2 + P))
    KH = 3
    al = torch.softmax(torch.randn(T, B, KH, dtype=x.dtype, device=dev), -1)
    gal = torch.randn(T, B, KH, dtype=x.dtype, device=dev) * 1e-3
    wh = torch.randn(KH, H, dtype=x.dtype, device=dev) * 0.1
    hpart = new(lib.kvae_lstm128_blocks(B), KH * (H + 1))
    wih0, whh0, whh1 = w("weight_ih_l0"), w("weight_hh_l0"), w("weight_hh_l1")
    whh0t, whh1t = whh0.t().contiguous(), whh1.t().contiguous()
    b0 = (w("bias_ih_l0") + w("bias_hh_l0")).contiguous()
    st = capi.current_stream_ptr()
    calls = {
        "lstm_layer0_forward": lambda: capi.check(lib.kvae_lstm128_forward(
            T, B, P, x.data_ptr(), wih0.data_ptr(), b0.data_ptr(), whh0t.data_ptr(),
            gates.data_ptr(), h.data_ptr(), c.data_ptr(), None, None, 1, st)),
        "lstm_layer1_forward": lambda: capi.check(lib.kvae_lstm128_forward(
            T, B, 0, None, None, None, whh1t.data_ptr(), gx.data_ptr(), h.data_ptr(), c.data_ptr(),
            None, None, 1, st)),
        # the upper layer's backward with the softmax head's adjoint in the kernel, as the
        # step runs it
        "lstm_layer1_backward": lambda: capi.check(lib.kvae_lstm128_backward_head(
            T, B, KH, gates.data_ptr(), c.data_ptr(), None, al.data_ptr(), gal.data_ptr(),
            wh.data_ptr(), whh1.data_ptr(), dG.data_ptr(), part.data_ptr(), hpart.data_ptr(), st)),
        "lstm_layer0_backward": lambda: capi.check(lib.kvae_lstm128_backward(
            T, B, P, x.data_ptr(), wih0.data_ptr(), gates.data_ptr(), c.data_ptr(), None,
            gh.data_ptr(), whh0.data_ptr(), dG.data_ptr(), dx.data_ptr(), part.data_ptr(), st)),
    }
    split = lib.kvae_lstm128_get_precision() == 1
    fam = "split" if split else "resident"
    kernel_of = {"lstm_layer0_forward": f"lstm128_{fam}_forward_kernel<{P}>",
                 "lstm_layer1_forward": f"lstm128_{fam}_forward_kernel<0>",
                 "lstm_layer1_backward": f"lstm128_{fam}_backward_kernel<0, {KH}>",
                 "lstm_layer0_backward": f"lstm128_{fam}_backward_kernel<{P}, 0>"}
    flops = 2.0 * T * B * 4 * H * H     # the recurrent product of one layer
    # compulsory HBM bytes of a launch: activated gates, h, c written (forward) / read with
    # the upstream gradient (c_{t-1} is carried from the previous step), gate gradients
    # written (backward)
    by = {"lstm_layer0_forward": 4.0 * T * B * (6 * H + P),
          "lstm_layer1_forward": 4.0 * T * B * (6 * H + 4 * H),
          "lstm_layer1_backward": 4.0 * T * B * (5 * H + 4 * H + 2 * KH),
          "lstm_layer0_backward": 4.0 * T * B * (6 * H + 4 * H + 2 * P)}
    if split:
        # the products around the recurrences, same arithmetic, each its own launch
        R = T * B
        wih1 = w("weight_ih_l1")
        wih1t = wih1.t().contiguous()
        b1 = (w("bias_ih_l1") + w("bias_hh_l1")).contiguous()
        h2, g2, d2 = h.view(R, H), gx.view(R, 4 * H), dG.view(R, 4 * H)
        gh2 = torch.empty(R, H, dtype=x.dtype, device=dev)
        S = lib.kvae_lstm128_wgrad_slabs(R)
        slabs = new(S, 4 * H * H)
        calls.update({
            "lstm_layer1_input_projection": lambda: capi.check(lib.kvae_rowstream_matmul(
                R, H, 4 * H, h2.data_ptr(), wih1t.data_ptr(), b1.data_ptr(), g2.data_ptr(), st)),
            "lstm_gradient_to_layer0": lambda: capi.check(lib.kvae_rowstream_matmul(
                R, 4 * H, H, d2.data_ptr(), wih1.data_ptr(), None, gh2.data_ptr(), st)),
            "lstm_weight_gradient": lambda: capi.check(lib.kvae_lstm128_wgrad(
                R, d2.data_ptr(), h2.data_ptr(), B, slabs.data_ptr(), 4 * H * H, st)),
        })
        kernel_of.update({"lstm_layer1_input_projection": "rowstream_matmul_kernel<128, 512>",
                          "lstm_gradient_to_layer0": "rowstream_matmul_kernel<512, 128>",
                          "lstm_weight_gradient": "wgrad_tn_kernel (one of three per step)"})
        by.update({"lstm_layer1_input_projection": 4.0 * R * 5 * H,
                   "lstm_gradient_to_layer0": 4.0 * R * 5 * H,
                   "lstm_weight_gradient": 4.0 * (R * 5 * H + S * 4 * H * H)})
    out = {}
    for name, fn in calls.items():
        sec = _timed(fn, iters)
        tf = flops / sec / 1e12
        if split:
            # three bf16 matrix instructions per product leave these kernels bound by the
            # bytes they stream; the fp32-equivalent matrix rate is listed beside it
            out[name] = {"kernel": kernel_of[name], "us": sec * 1e6, "bound": "hbm",
                         "achieved": by[name] / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by[name] / sec / 1e9 / HBM_PEAK_GBS, "flops_per_launch": flops,
                         "hbm_bytes_per_launch": by[name], "fp32_equivalent_TFLOPs": tf,
                         "vs_fp32_matrix_peak": tf / MFMA_FP32_PEAK_TF}
        else:
            out[name] = {"kernel": kernel_of[name], "us": sec * 1e6, "bound": "mfma",
                         "achieved": tf, "peak": MFMA_FP32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": tf / MFMA_FP32_PEAK_TF, "flops_per_launch": flops,
                         "hbm_bytes_per_launch": by[name], "hbm_GBps": by[name] / sec / 1e9}
    return out


def time_fused_kernels(ssm, a, burn_in, iters=30):
    """HIP-event durations of the fused objective's two launches (kvae_kalman_loss_forward /
    _backward: filter -> smoother -> ELBO terms and their adjoint) through the C ABI on the
    tensors of the bench, with the weight network's real mixing weights."""
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd import functions as F
    lib = capi.load()
    dev, dt = a.device, a.dtype
    T, B, p = a.shape
    n, K = ssm.z_dim, ssm.K
    if not lib.kvae_loss_supported(n, p, K, capi.dtype_code(dt)):
        return None
    with torch.no_grad():
        ssm.weight_model.clear_hidden_state()
        alpha = ssm.weight_model(a, lead=True).contiguous()
        A_K, C_K = ssm.mat_A_K.contiguous(), ssm.mat_C_K.contiguous()
        Q, R = ssm.mat_Q.contiguous(), ssm.mat_R.contiguous()
        m0, P0 = ssm.initial_state_mean, ssm.initial_state_covariance
    new = lambda *s: torch.empty(s, dtype=dt, device=dev)
    dims = capi.Dims(T, B, n, p, K, capi.dtype_code(dt), 1, burn_in)
    saved = torch.empty(lib.kvae_loss_saved_bytes(C.byref(dims)), dtype=torch.uint8, device=dev)
    eps, out = torch.randn(T, B, n, dtype=dt, device=dev), new(4)
    lf = capi.fill(capi.LossFwd(), a=a.detach(), alpha=alpha, mask=None, A_K=A_K, C_K=C_K, Q=Q,
                   R=R, m0=m0, P0=P0, eps=eps, saved=saved, terms=None, out=out,
                   status=F.status_word(dev))
    c = 1.0 / (T * B)
    lf.w_obs, lf.w_trans, lf.w_post = c, c, -c
    d = dict(a=new(T, B, p), alpha=new(T + 1, B, K), A_K=new(K, n, n), C_K=new(K, p, n),
             Q=new(n, n), R=new(p, p))
    scale = torch.full((1,), -1.0, dtype=dt, device=dev)
    lb = capi.LossBwd()
    lb.fwd = lf
    capi.fill(lb, g_scale=scale, g_scale_stride=0, d_a=d["a"], d_alpha=d["alpha"], d_A_K=d["A_K"],
              d_C_K=d["C_K"], d_Q=d["Q"], d_R=d["R"], d_m0=None, d_P0=None)
    nf = lib.kvae_workspace_bytes(C.byref(dims), capi.OP_LOSS_FWD)
    nb = lib.kvae_workspace_bytes(C.byref(dims), capi.OP_LOSS_BWD)
    nv = lib.kvae_workspace_bytes(C.byref(dims), capi.OP_LOSS_VALUE_AND_GRAD)
    ws = torch.empty(max(nf, nb, nv, 8), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    tf = _timed(lambda: capi.check(lib.kvae_kalman_loss_forward(
        C.byref(dims), C.byref(lf), ws.data_ptr(), nf, stream)), iters)
    tb = _timed(lambda: capi.check(lib.kvae_kalman_loss_backward(
        C.byref(dims), C.byref(lb), ws.data_ptr(), nb, stream)), iters)
    # what the training step calls: value and gradient from one call
    tv = _timed(lambda: capi.check(lib.kvae_kalman_loss_value_and_grad(
        C.byref(dims), C.byref(lb), ws.data_ptr(), nv, stream)), iters)
    on_chip = lib.kvae_loss_saved_bytes(C.byref(dims)) <= 256   # the 16-lanes-per-sequence form
    return {"loss_value_and_grad": tv, "loss_forward": tf, "loss_backward": tb}, on_chip


def fused_bytes(n, p, K, el, on_chip=False):
    """Compulsory global-memory bytes per sequence-step of the fused objective's calls.
    With the per-step state on chip (16 lanes per sequence, LDS): the inputs (a, alpha, eps)
    read and d_a, d_alpha written, nothing else.  Otherwise also the state handed from the
    forward to the backward pass (filtered, predicted and smoothed means and covariances)
    written once and read once and the gradients handed from the backward pass's first sweep
    to its second written and read."""
    v, c = n, n * n
    inputs = p + K + n
    if on_chip:
        return {"loss_forward": el * inputs, "loss_backward": el * (inputs + p + K),
                "loss_value_and_grad": el * (inputs + p + K)}
    saved = 3 * v + 3 * c
    bars = 2 * v + 2 * c
    fwd = el * (inputs + saved + (2 * v + 2 * c))                      # + the filter state re-read
    bwd = el * (inputs + saved + (2 * v + 2 * c) + 2 * bars + 2 * (p + K))
    return {"loss_forward": fwd, "loss_backward": bwd, "loss_value_and_grad": fwd + bwd}


def time_kernels(ssm, a, burn_in, iters=30, synthetic_alpha=False):
    """Average duration of each HIP kernel of the step, measured with HIP events on
    the stream the kernels are launched on, by replaying each launch `iters` times
    on the tensors of a real forward/backward."""
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd import functions as F
    lib = capi.load()
    dev = a.device
    T, B, p = a.shape
    n, K = ssm.z_dim, ssm.K
    dt = a.dtype
    with torch.no_grad():
        if synthetic_alpha:
            alpha = torch.softmax(torch.randn(T + 1, B, K, dtype=dt, device=dev), -1).contiguous()
        else:
            ssm.weight_model.clear_hidden_state()
            alpha = torch.cat([torch.full((1, B, K), 1.0 / K, dtype=dt, device=dev),
                               ssm.weight_model(a)]).contiguous()
        A_K, C_K = ssm.mat_A_K.contiguous(), ssm.mat_C_K.contiguous()
        Q, R = ssm.mat_Q.contiguous(), ssm.mat_R.contiguous()
        LR = torch.linalg.cholesky(R).contiguous()
        m0, P0 = ssm.initial_state_mean, ssm.initial_state_covariance
    new = lambda *s: torch.empty(s, dtype=dt, device=dev)
    rnd = lambda *s: torch.randn(s, dtype=dt, device=dev)
    dims = capi.Dims(T, B, n, p, K, capi.dtype_code(dt), 1, burn_in)
    st = F.status_word(dev)
    t = dict(m=new(T, B, n), P=new(T, B, n, n), mn=new(T, B, n), Pn=new(T, B, n, n),
             As=new(T + 1, B, n, n), Cs=new(T + 1, B, p, n), ms=new(T, B, n), Ps=new(T, B, n, n),
             zs=new(T, B, n), ares=new(T, B, p), obs=new(T, B), trans=new(T, B), post=new(T, B),
             z=new(T, B, n), es=rnd(T, B, n), eo=rnd(T, B, p), ee=rnd(T, B, n))
    ff = capi.fill(capi.FilterFwd(), a=a.detach(), alpha=alpha, mask=None, A_K=A_K, C_K=C_K, Q=Q,
                   R=R, m_in=m0, P_in=P0, carry_per_sequence=0, mean_filt=t["m"], cov_filt=t["P"],
                   mean_pred=t["mn"], cov_pred=t["Pn"], mat_As=t["As"], mat_Cs=t["Cs"], status=st)
    sf = capi.fill(capi.SmoothFwd(), mean_filt=t["m"], cov_filt=t["P"], mean_pred=t["mn"],
                   cov_pred=t["Pn"], mat_As=t["As"], mat_Cs=t["Cs"], LR=LR, eps_state=t["es"],
                   eps_obs=t["eo"], mean_smooth=t["ms"], cov_smooth=t["Ps"], zs=t["zs"],
                   as_resampled=t["ares"], status=st)
    ef = capi.fill(capi.ElboFwd(), a=a.detach(), mean_smooth=t["ms"], cov_smooth=t["Ps"],
                   mat_As=t["As"], mat_Cs=t["Cs"], Q=Q, R=R, m0=m0, P0=P0, eps=t["ee"],
                   obs=t["obs"], trans=t["trans"], post=t["post"], z=t["z"], status=st)
    g = dict(m=rnd(T, B, n), P=rnd(T, B, n, n), mn=rnd(T, B, n), Pn=rnd(T, B, n, n),
             As=rnd(T + 1, B, n, n), Cs=rnd(T + 1, B, p, n), ms=rnd(T, B, n), Ps=rnd(T, B, n, n),
             obs=rnd(T, B), trans=rnd(T, B), post=rnd(T, B))
    d = dict(a=new(T, B, p), alpha=new(T + 1, B, K), m_in=new(B, n), P_in=new(B, n, n),
             A_K=new(K, n, n), C_K=new(K, p, n), Q=new(n, n), R=new(p, p), m=new(T, B, n),
             P=new(T, B, n, n), mn=new(T, B, n), Pn=new(T, B, n, n), As=new(T + 1, B, n, n),
             Cs=new(T + 1, B, p, n), ms=new(T, B, n), Ps=new(T, B, n, n), m0=new(n), P0=new(n, n))
    fb = capi.FilterBwd()
    fb.fwd = ff
    capi.fill(fb, g_mean_filt=g["m"], g_cov_filt=g["P"], g_mean_pred=g["mn"], g_cov_pred=g["Pn"],
              g_mat_As=g["As"], g_mat_Cs=g["Cs"], d_a=d["a"], d_alpha=d["alpha"], d_m_in=d["m_in"],
              d_P_in=d["P_in"], d_A_K=d["A_K"], d_C_K=d["C_K"], d_Q=d["Q"], d_R=d["R"])
    sb = capi.SmoothBwd()
    sb.fwd = sf
    capi.fill(sb, g_mean_smooth=g["ms"], g_cov_smooth=g["Ps"], g_zs=None, g_as_resampled=None,
              d_mean_filt=d["m"], d_cov_filt=d["P"], d_mean_pred=d["mn"], d_cov_pred=d["Pn"],
              d_mat_As=d["As"], d_mat_Cs=d["Cs"], d_LR=None)
    eb = capi.ElboBwd()
    eb.fwd = ef
    capi.fill(eb, g_obs=g["obs"], g_trans=g["trans"], g_post=g["post"], d_a=d["a"],
              d_mean_smooth=d["ms"], d_cov_smooth=d["Ps"], d_mat_As=d["As"], d_mat_Cs=d["Cs"],
              d_Q=d["Q"], d_R=d["R"], d_m0=d["m0"], d_P0=d["P0"])
    ws_bytes = max(lib.kvae_workspace_bytes(C.byref(dims), op) for op in (1, 2, 3))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    calls = {
        "filter_forward": lambda: lib.kvae_filter_forward(C.byref(dims), C.byref(ff), stream),
        "smooth_forward": lambda: lib.kvae_smooth_forward(C.byref(dims), C.byref(sf), stream),
        "elbo_forward": lambda: lib.kvae_elbo_forward(C.byref(dims), C.byref(ef), stream),
        "elbo_backward": lambda: lib.kvae_elbo_backward(C.byref(dims), C.byref(eb), ws.data_ptr(),
                                                        ws_bytes, stream),
        "smooth_backward": lambda: lib.kvae_smooth_backward(C.byref(dims), C.byref(sb),
                                                            ws.data_ptr(), ws_bytes, stream),
        "filter_backward": lambda: lib.kvae_filter_backward(C.byref(dims), C.byref(fb),
                                                            ws.data_ptr(), ws_bytes, stream),
    }
    out = {}
    for name, fn in calls.items():   # forward kernels first: they fill what the others read
        for _ in range(3):
            assert fn() == 0, name
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) * 1e-3 / iters   # seconds per launch
    return out


def _progress(msg):
    """A line on stderr (long configurations must not look hung)."""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_time_objective(T, B, net, burn_in, n, p, K, threads, budget_s, max_iters, min_iters=3):
    """Median seconds of one forward+backward of the oracle's Kalman objective."""
    from kalman_vae_amd import StateSpaceModel
    from oracle import kalman_oracle as O
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    ssm = StateSpaceModel(a_dim=p, z_dim=n, K=K, dynamics_parameter_network=net)
    ssm.mat_Q = torch.eye(n)
    ssm.mat_R = torch.eye(p)
    ssm.train()
    a = bouncing_ball_observations(T, B, p, "cpu", seed=0).requires_grad_(True)

    def step():
        ssm.zero_grad(set_to_none=True)
        a.grad = None
        ssm.weight_model.clear_hidden_state()
        eps = torch.randn(T, B, n)
        fn = (torch.randn(T, B, n), torch.randn(T, B, p))
        sn = (torch.randn(T, B, n), torch.randn(T, B, p))
        obj, _ = O.kalman_objective(
            a, ssm.mat_A_K, ssm.mat_C_K, ssm.mat_Q, ssm.mat_R, ssm.initial_state_mean,
            ssm.initial_state_covariance, ssm.weight_model, eps, filter_noise=fn,
            smooth_noise=sn, burn_in=burn_in)
        (-obj).backward()

    step()  # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < min_iters or (time.perf_counter() - t_start < budget_s
                                     and len(times) < max_iters):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2], len(times)


def cpu_baseline(T, B, net, burn_in, n=4, p=2, K=3, budget_s=12.0):
    """The oracle (per-step PyTorch ops, autograd backward) on the host cores: the
    workload's own batch with all threads (`value`), and beside it the same batch on one
    thread and the reference's own batch of 32 sequences with all threads and with one."""
    all_threads = torch.get_num_threads()
    unit = "sequence-steps/s"
    full_B, full_T = B, T
    if T * B > 100000:
        # a bounded sample of the workload: fewer and shorter sequences of the same shape
        # (the oracle's cost per sequence-step is what is measured; 1000 steps of 16
        # sequences at z_dim=32 take over a minute per iteration)
        T = min(T, 100)
        B = max(8, 1600 // T)
    _progress(f"cpu baseline: oracle at T={T} B={B}, {all_threads} threads")
    med, iters = _cpu_time_objective(T, B, net, burn_in, n, p, K, all_threads, budget_s, 30,
                                     min_iters=3 if full_T <= 100 else 2)
    out = {"value": T * B / med, "unit": unit, "cores": all_threads, "kind": "port",
           "sample": f"{iters} fwd+bwd iterations of T={T} B={B}"
                     + (f" (a sample of the workload's T={full_T} B={full_B})"
                        if (B, T) != (full_B, full_T) else "")
                     + f" (median {med*1e3:.1f} ms), "
                     f"oracle/kalman_oracle.py with the {net} weight network, torch "
                     f"{torch.__version__} CPU, {os.cpu_count()} logical CPUs visible",
           "cpu_model": _cpu_model(), "also": []}
    extra = [(B, 1, 6.0, 3)] if full_T * full_B <= 200000 else []
    if B != 32 and full_T <= 50:
        extra += [(32, all_threads, 3.0, 30), (32, 1, 3.0, 30)]
    for b, threads, budget, cap in extra:
        _progress(f"cpu baseline: oracle at T={T} B={b}, {threads} thread(s)")
        m, it = _cpu_time_objective(T, b, net, burn_in, n, p, K, threads, budget, cap)
        out["also"].append({"batch": b, "cores": threads, "value": T * b / m, "unit": unit,
                            "ms_per_iteration": m * 1e3, "iterations": it})
    torch.set_num_threads(all_threads)
    return out


CONFIGS = {
    # name: (T, B per GPU, p, n, K, steps, warmup)
    "c1": dict(seq_len=20, batch=32, obs_dim=2, state_dim=4, bases=3),
    "c2": dict(seq_len=20, batch=4096, obs_dim=2, state_dim=4, bases=3),
    "c3": dict(seq_len=20, batch=1048576, obs_dim=2, state_dim=4, bases=3),
    "c4": dict(seq_len=1000, batch=1024, obs_dim=16, state_dim=32, bases=8),
    "c5": dict(seq_len=20, batch=128, obs_dim=2, state_dim=4, bases=3),
}


def _dist_setup(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # KVAE_BENCH_REHEARSAL=1: rehearse the multi-rank code path on a box with ONE GPU
    # (every rank on cuda:0, gloo instead of RCCL); the numbers mean nothing then.
    rehearsal = os.environ.get("KVAE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return rank, world, dev, rehearsal


def _fence(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(elapsed, world, dev, rehearsal):
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


# Untimed steps of set-up, before the W warm-up steps, for the configurations whose step is
# shorter than a millisecond: a fresh process reaches its steady clock only after about 15 ms of
# back-to-back work (c2 with --steps 20: 0.735 ms per step after 5 warm-up steps, 0.699 ms after
# 30 or 100, 0.698 ms over 200 timed steps).  The same count on every rank (a step may hold a
# collective); reported as `settle_steps`.
SETTLE_STEPS = {"c1": 40, "c2": 40, "c3": 0, "c4": 0, "c5": 4}


def _timed_steps(step, args, world, dev, rehearsal):
    for _ in range(SETTLE_STEPS.get(args.config, 0)):
        step()
    for _ in range(args.warmup):
        step()
    _fence(world)
    _progress(f"{args.warmup} warm-up steps done, timing {args.steps} steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _fence(world)
    return _max_over_ranks(time.perf_counter() - t0, world, dev, rehearsal)


def _base_result(args, world, units_per_step, elapsed, scaling, workload, extra_cfg):
    return {
        "metric": "filter+smoother sequence-steps/sec (fwd+bwd)", "value":
        units_per_step * args.steps / elapsed, "unit": "sequence-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup,
        "settle_steps": SETTLE_STEPS.get(args.config, 0), "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "config": dict({"workload": workload}, **extra_cfg)}


# -- c3: masked imputation, forward only, the batch split over the ranks ----------------------
def run_imputation(args, rank, world, dev, rehearsal):
    from kalman_vae_amd import SampleControl
    from kalman_vae_amd.parallel import shard_bounds
    T, p, n, K = args.seq_len, args.obs_dim, args.state_dim, args.bases
    total = args.batch
    begin, end = shard_bounds(total, world, rank)
    mine = end - begin
    chunk = min(args.chunk, mine)
    torch.manual_seed(0)
    ssm = build_model(dev, torch.float32, args.net, n, p, K).eval()
    sc = SampleControl(encoder="mean", decoder="mean", state_transition="mean", observation="mean")
    # one chunk of synthetic observations, reused for every chunk of the shard (the kernels'
    # time does not depend on the values); resident in HBM before the timed region
    a = bouncing_ball_observations(T, chunk, p, dev, seed=2000 + rank).to(torch.float32)
    g = torch.Generator(device="cpu").manual_seed(3000 + rank)
    mask = (torch.rand(T, chunk, generator=g) < 0.7).float()
    mask[0] = 1.0
    mask[-1] = 1.0
    mask = mask.to(dev)
    n_chunks = (mine + chunk - 1) // chunk

    fused = not args.separate_kernels

    def one_chunk():
        if fused:       # filter, weight network and smoother in one launch, two tensors written
            return ssm.impute(a, sc, observation_mask=mask)
        out = ssm.kalman_filter(a, sc, observation_mask=mask)
        return out[6], ssm.kalman_smooth(a, *out[:6], sample_control=sc)[3]

    def step():
        with torch.no_grad():
            for _ in range(n_chunks):
                one_chunk()

    elapsed = _timed_steps(step, args, world, dev, rehearsal)
    if rank != 0:
        return None
    units = T * n_chunks * chunk * world
    res = _base_result(
        args, world, units, elapsed, "strong",
        f"masked imputation (filter + RTS smoother, forward only, keep probability 0.7, first "
        f"and last step observed), T={T} a_dim={p} z_dim={n} K={K}, {total} sequences over "
        f"{world} GPU(s) in chunks of {chunk}, {args.net} weight network",
        {"seq_len": T, "total_sequences": total, "chunk": chunk, "parallelism": f"shard{world}",
         "collectives": "none"})
    res["metric"] = "filter+smoother sequence-steps/sec (masked imputation, forward)"
    res["config"]["one_launch_per_chunk"] = bool(fused and args.net == "mlp")
    if not args.no_kernel_timing:
        el = 4
        with torch.no_grad():
            if fused and args.net == "mlp":
                # one launch per chunk: reads a, mask; writes a_w, alpha, as_resampled; the
                # filtered moments go through the workspace once each way
                tc = _timed(one_chunk, 5)
                by = el * ((p + 1) + 2 * p + 2 * K + 2 * (n + n * n)) * T * chunk
                res["kernels"] = {"filter_smooth_fused": {"us": tc * 1e6, "GBps": by / tc / 1e9}}
                dom, sec = "filter_smooth_fused (one chunk, host glue included)", tc
            else:
                tf = _timed(lambda: ssm.kalman_filter(a, sc, observation_mask=mask), 5)
                out = ssm.kalman_filter(a, sc, observation_mask=mask)
                ts = _timed(lambda: ssm.kalman_smooth(a, *out[:6], sample_control=sc), 5)
                fb = el * ((p + 1) + (2 * n + 2 * n * n + n * n + p * n + p + K)) * T * chunk
                sb = kernel_bytes(n, p, K, el)["smooth_forward"] * T * chunk
                dom, sec, by = ("masked filter", tf, fb) if tf >= ts else ("smoother", ts, sb)
                dom += " stage of one chunk"
                res["kernels"] = {"masked_filter": {"us": tf * 1e6, "GBps": fb / tf / 1e9},
                                  "smoother": {"us": ts * 1e6, "GBps": sb / ts / 1e9}}
        traffic, table = None, {}
        try:      # HBM bytes per launch counted by the PMC passes committed under profiles/
            with open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")) as fh:
                table = json.load(fh)
            traffic = table.get(f"T{T}_B{chunk}_n{n}_p{p}_K{K}", {}).get(
                "filter_smooth_fused" if fused and args.net == "mlp" else "")
        except OSError:
            pass
        res["roofline"] = {"kernel": dom, "bound": "hbm",
                           "achieved": by / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": by / sec / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                           "algorithmic_bytes_per_launch": by, "avg_launch_us": sec * 1e6}
        step = None
        if args.net == "lstm":
            # operands as the workload has them: the state the network carries after a chunk
            # (the clock the chip holds, and so the duration, depends on the values)
            with torch.no_grad():
                one_chunk()
            step = time_lstm_step_kernels(ssm, chunk, a[-1], ssm.weight_model.hidden)
        if step is not None:
            # with the LSTM network the workload is T steps of its two layers per chunk, one
            # launch per layer and step; the upper layer's launch is the dominant kernel
            res["kernels"].update(step)
            up = step["lstm_step_upper_layer"]
            res["kalman_roofline"] = res["roofline"]
            res["roofline"] = {
                "kernel": "lstm128_step_fused_kernel<0> (upper LSTM layer, one time step of one "
                          "chunk, input projection in the kernel)",
                "bound": "mfma", "achieved": up["TFLOPs_bf16"], "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": up["TFLOPs_bf16"] / MFMA_BF16_PEAK_TFLOPS,
                "traffic": table.get(f"lstm_step_B{chunk}", {}).get("upper_layer"),
                "flops_per_launch": up["flops_per_launch"], "avg_launch_us": up["us"],
                "hbm_GBps": up["GBps"],
                "share_of_step": up["us"] * 1e-6 * T * n_chunks / (elapsed / args.steps),
                "note": "three bf16 matrix instructions per fp32 product (hi/lo split), counted "
                        "as executed; 2.5 KB per row through HBM"}
    return res


def time_lstm_step_kernels(ssm, rows, x, hidden, iters=20):
    """HIP-event durations of the two launches that step the default LSTM network (two layers
    of 128 units) once over ``rows`` sequences, as the masked filter issues them per time step,
    from the carried state ``hidden`` = (h (2,rows,128), c (2,rows,128)) on the input x
    (rows, a_dim); None for any other network."""
    lstm = getattr(ssm.weight_model, "lstm", None)
    if (lstm is None or lstm.num_layers != 2 or lstm.hidden_size != 128 or lstm.input_size > 2
            or hidden is None):
        return None
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd.lstm_function import _lstm2_operands
    lib, H, P = capi.load(), 128, lstm.input_size
    dev = lstm.weight_hh_l0.device
    w = lambda name: getattr(lstm, name).detach()
    whh0t, wih1t, whh1t, b0, b1 = _lstm2_operands(
        w("weight_hh_l0"), w("bias_ih_l0"), w("bias_hh_l0"), w("weight_ih_l1"), w("weight_hh_l1"),
        w("bias_ih_l1"), w("bias_hh_l1"))
    wih0 = w("weight_ih_l0").contiguous()
    x, hp, cp = x.contiguous(), hidden[0].contiguous(), hidden[1].contiguous()
    h, c = torch.empty(2, rows, H, device=dev), torch.empty(2, rows, H, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def lower():
        capi.check(lib.kvae_lstm128_step_fused(
            rows, P, x.data_ptr(), wih0.data_ptr(), hp[0].data_ptr(), cp[0].data_ptr(),
            whh0t.data_ptr(), b0.data_ptr(), h[0].data_ptr(), c[0].data_ptr(), st))

    def upper():
        capi.check(lib.kvae_lstm128_step_fused(
            rows, 0, h[0].data_ptr(), wih1t.data_ptr(), hp[1].data_ptr(), cp[1].data_ptr(),
            whh1t.data_ptr(), b1.data_ptr(), h[1].data_ptr(), c[1].data_ptr(), st))

    out = {}
    for name, fn, depth, by in (("lstm_step_lower_layer", lower, H, 4 * (4 * H + P)),
                                ("lstm_step_upper_layer", upper, 2 * H, 4 * 5 * H)):
        t = _timed(fn, iters)
        flops = 3 * 2 * depth * 4 * H * rows          # three bf16 products per fp32 product
        out[name] = {"us": t * 1e6, "flops_per_launch": flops, "TFLOPs_bf16": flops / t / 1e12,
                     "GBps": by * rows / t / 1e9, "rows": rows}
    return out


# -- c5: the full model, conv encoder / decoder on 64x64 frames ----------------------------------
def run_full_model(args, rank, world, dev, rehearsal):
    from kalman_vae_amd import KalmanVariationalAutoencoder, SampleControl
    from kalman_vae_amd.parallel import DataParallelStep
    T, B, image = args.seq_len, args.batch, args.image
    torch.manual_seed(1234)          # identical initial parameters on every rank
    model = KalmanVariationalAutoencoder(
        image_size=(image, image), image_channels=1, a_dim=args.obs_dim, z_dim=args.state_dim,
        K=args.bases, init_transition_reg_weight=0.9, init_observation_reg_weight=0.9,
        learn_noise_covariance=False, init_noise_scale=1.0,
        dynamics_parameter_network=args.net, decoder_type="bernoulli").to(dev).train()
    model.state_space_model.fused_loss = not args.separate_kernels
    model.state_space_model.rng_parity = False
    model.state_space_model.check_numerics = False
    dp = DataParallelStep(model, torch.optim.Adam(model.parameters(), lr=1e-3))
    torch.manual_seed(99 + rank)     # every rank its own synthetic frames
    # binary discs of radius 4 px at the positions of a bouncing ball
    pos = bouncing_ball_observations(T, B, 2, "cpu", seed=500 + rank)
    yy, xx = torch.meshgrid(torch.arange(image), torch.arange(image), indexing="ij")
    cx = ((pos[..., 0] + 1) * 0.5 * (image - 1)).view(T, B, 1, 1)
    cy = ((pos[..., 1] + 1) * 0.5 * (image - 1)).view(T, B, 1, 1)
    xs = (((xx - cx) ** 2 + (yy - cy) ** 2) <= 16.0).float().unsqueeze(2).to(dev)
    sc = SampleControl()

    def step():
        dp(lambda: model.elbo(sc, xs=xs, burn_in=args.burn_in)[0])

    elapsed = _timed_steps(step, args, world, dev, rehearsal)
    if rank != 0:
        return None
    res = _base_result(
        args, world, T * B * world, elapsed, "weak",
        f"full Kalman-VAE training step (conv encoder/decoder, Kalman objective, backward, "
        f"one flat gradient all-reduce, Adam), {image}x{image} frames, T={T}, {B} sequences per "
        f"GPU, {args.net} weight network",
        {"seq_len": T, "batch_per_gpu": B, "global_batch": B * world, "image": image,
         "parallelism": f"dp{world}", "fused_objective": not args.separate_kernels})
    res["metric"] = "full KVAE training step, sequence-steps/sec (= frames/sec)"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="benchmark configuration (see the module docstring); sets the shape "
                         "arguments below unless they are given")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (c3: in total)")
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--net", default=None, choices=["lstm", "mlp"])
    ap.add_argument("--burn-in", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fp32-network", action="store_true",
                    help="skip the second timed run with exact fp32 weight-network products")
    ap.add_argument("--state-dim", type=int, default=None)
    ap.add_argument("--obs-dim", type=int, default=None)
    ap.add_argument("--bases", type=int, default=None)
    # (a GPU has 288 GB: a shard's million sequences go through in one call - 34.1 ms in chunks
    # of 262 144, 32.8 ms in chunks of 524 288, 32.4 ms in one; smaller chunks are slower still)
    ap.add_argument("--chunk", type=int, default=1048576, help="c3: sequences per call")
    ap.add_argument("--image", type=int, default=64, help="c5: frame height and width")
    ap.add_argument("--kernels-only", action="store_true",
                    help="time only the six separate HIP kernels with synthetic mixing weights "
                         "(softmax of N(0,1)); no PyTorch weight network, no autograd glue")
    ap.add_argument("--kernel-iters", type=int, default=30)
    ap.add_argument("--separate-kernels", action="store_true",
                    help="run filter, smoother and ELBO terms as six separate kernels with all "
                         "their tensors materialised (the drop-in API's path) instead of the "
                         "fused objective")
    ap.add_argument("--no-hip-graph", action="store_true",
                    help="run the step eagerly instead of replaying it as one HIP graph")
    args = ap.parse_args()
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.net is None:
        # the reference's default network everywhere (--net mlp: the two-layer perceptron that
        # the one-launch masked filter of c3 evaluates inside the kernel)
        args.net = "lstm"
    defaults = {"c1": (50, 10), "c2": (50, 10), "c3": (5, 2), "c4": (3, 1), "c5": (10, 3)}
    if args.steps is None:
        args.steps = defaults[args.config][0]
    if args.warmup is None:
        args.warmup = defaults[args.config][1]

    rank, world, dev, rehearsal = _dist_setup(args)
    result = None
    if args.config == "c3":
        result = run_imputation(args, rank, world, dev, rehearsal)
    elif args.config == "c5":
        result = run_full_model(args, rank, world, dev, rehearsal)
    else:
        result = run_training(args, rank, world, dev, rehearsal)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        result["config"]["name"] = args.config
        print(json.dumps(result))


def run_training(args, rank, world, dev, rehearsal):
    from kalman_vae_amd import SampleControl, kalman_elbo
    from kalman_vae_amd import functions as F
    from kalman_vae_amd.parallel import FlatGradBucket

    T, B = args.seq_len, args.batch
    dtype = torch.float32
    torch.manual_seed(0)   # same initial parameters on every rank
    n, p, K = args.state_dim, args.obs_dim, args.bases
    ssm = build_model(dev, dtype, args.net, n, p, K, fused=not args.separate_kernels)
    fused = ssm.fused_loss and F.loss_supported(n, p, K, dtype)
    a = bouncing_ball_observations(T, B, p, dev, seed=1000 + rank).to(dtype).requires_grad_(True)
    bucket = FlatGradBucket(ssm.parameters()) if world > 1 else None
    sc = SampleControl()

    def reduce_gradients():
        # asynchronous on the collective's own stream: it overlaps the next replay, which
        # reads no gradient; the next call waits for it before it refills the bucket
        if bucket is not None:
            bucket.all_reduce(average=True, async_op=not rehearsal)

    def eager_step():
        ssm.zero_grad(set_to_none=True)
        a.grad = None
        obj, _ = kalman_elbo(ssm, a, sc, burn_in=args.burn_in)
        (-obj).backward()
        reduce_gradients()

    # The whole step (weight network, filter, smoother, ELBO terms and their backward)
    # replayed as one HIP graph; same kernels, same work.
    graphed, graph_note = None, None
    if not args.no_hip_graph and not args.kernels_only and T <= 100:
        from kalman_vae_amd.graphed import GraphedElboStep
        try:
            graphed = GraphedElboStep(ssm, a, sc, burn_in=args.burn_in)
        except Exception as err:   # capture refused or failed: time the eager step instead
            graph_note = f"{type(err).__name__}: {err}"
            torch.cuda.synchronize()

    def graph_step():
        graphed(None)            # the synthetic batch already is the static input
        reduce_gradients()

    step = graph_step if graphed is not None else eager_step

    if args.kernels_only:
        kt = time_kernels(ssm, a, args.burn_in, iters=args.kernel_iters, synthetic_alpha=True)
        elapsed = sum(kt.values()) * args.steps
    else:
        elapsed = _timed_steps(step, args, world, dev, rehearsal)
        if bucket is not None:
            bucket.wait()
            bucket.check()
    if rank != 0:
        return None
    units = T * B
    result = _base_result(
        args, world, world * units, elapsed, "weak",
        f"bouncing-ball shapes T={T} a_dim={p} z_dim={n} K={K}, {B} sequences per GPU, "
        + ("synthetic mixing weights, six separate HIP kernels only, " if args.kernels_only
           else f"{args.net} weight network, ")
        + "filter+RTS smoother+ELBO terms fwd+bwd"
        + (" (fused objective: value and gradient in one call)" if fused and not args.kernels_only
           else " (six separate kernels)"),
        {"seq_len": T, "batch_per_gpu": B, "global_batch": B * world, "burn_in": args.burn_in,
         "parallelism": f"dp{world}" + (" (rehearsal: one GPU, gloo)" if rehearsal else ""),
         "hip_graph": graphed is not None, "fused_objective": bool(fused)})
    if graph_note:
        result["config"]["hip_graph_note"] = graph_note
    if args.net == "lstm" and not args.kernels_only:
        # the Kalman path is fp32 throughout; the weight network's matrix products (78 % of the
        # step) are not: say so, and time the same step with exact fp32 products beside it
        from kalman_vae_amd import _capi as capi
        lib = capi.load()
        split = lib.kvae_lstm128_get_precision() == 1
        result["config"]["weight_network_products"] = (
            "bf16x3-split (fp32 operands as hi+lo bf16, three bf16 matrix instructions per "
            "product, fp32 accumulation; about 1e-5 relative per product)" if split
            else "fp32 matrix instructions")
        result["config"]["kalman_path_arithmetic"] = "fp32"
        if split:
            result["config"]["workload"] += "; weight network products bf16x3-split"
        if split and world == 1 and T <= 100 and not args.no_fp32_network:
            _progress("timing the same step with exact fp32 weight-network products")
            lib.kvae_lstm128_set_precision(0)
            try:
                exact = None
                if graphed is not None:
                    from kalman_vae_amd.graphed import GraphedElboStep
                    g32 = GraphedElboStep(ssm, a, sc, burn_in=args.burn_in)
                    exact = lambda: g32(None)
                else:
                    exact = eager_step
                e32 = _timed_steps(exact, args, world, dev, rehearsal)
                result["fp32_network"] = {
                    "ms_per_step": e32 / args.steps * 1e3, "value": units * args.steps / e32,
                    "unit": "sequence-steps/s",
                    "note": "same step, same process, kvae_lstm128_set_precision(0): every "
                            "product of the weight network on fp32 matrix instructions"}
            finally:
                lib.kvae_lstm128_set_precision(1)
    step_s = elapsed / args.steps
    fl = flops_forward(n, p, K)
    step_flops = 3 * sum(fl.values()) * units
    # the whole timed step against the algorithmic work of the Kalman path alone
    result["valu_frac_step"] = step_flops / step_s / 1e12 / VALU_FP32_PEAK_TF
    if not args.no_kernel_timing:
        _progress("timing individual kernels")
        kernels = {}
        kal = None      # (name, seconds, flops, bytes) of the Kalman path's longest kernel
        if fused and not args.kernels_only:
            ft, on_chip = time_fused_kernels(ssm, a, args.burn_in, iters=args.kernel_iters)
            fb = fused_bytes(n, p, K, 4, on_chip)
            ffl = {"loss_forward": sum(fl.values()), "loss_backward": 2 * sum(fl.values()),
                   "loss_value_and_grad": 3 * sum(fl.values())}
            for name, sec in ft.items():
                kernels[name] = {"us": sec * 1e6, "GBps": fb[name] * units / sec / 1e9,
                                 "TFLOPs": ffl[name] * units / sec / 1e12}
            # the step calls the one-call form
            dom = "loss_value_and_grad"
            kal = (dom, ft[dom], ffl[dom] * units, fb[dom] * units)
            total_k = ft[dom]
            result["config"]["objective_state"] = (
                "on chip (LDS): 16 lanes per sequence, one launch for value and gradient"
                if on_chip else "through device memory: forward and backward launch")
            note = ("kvae_kalman_loss_value_and_grad (value, d_a, d_alpha and the parameter "
                    "gradients; with its row-sum launch) with the weight network's mixing "
                    "weights; excludes the weight network and the noise draw; loss_forward / "
                    "loss_backward are the two-call form of the same work, timed beside it")
        else:
            if not args.kernels_only:
                kt = time_kernels(ssm, a, args.burn_in, iters=args.kernel_iters)
            kb = kernel_bytes(n, p, K, 4)
            kflops = {"filter_forward": fl["filter"], "filter_backward": 2 * fl["filter"],
                      "smooth_forward": fl["smooth"], "smooth_backward": 2 * fl["smooth"],
                      "elbo_forward": fl["elbo"], "elbo_backward": 2 * fl["elbo"]}
            for name, sec in kt.items():
                kernels[name] = {"us": sec * 1e6, "GBps": kb[name] * units / sec / 1e9,
                                 "TFLOPs": kflops[name] * units / sec / 1e12}
            dom = max(kt, key=kt.get)
            kal = (dom, kt[dom], kflops[dom] * units, kb[dom] * units)
            total_k = sum(kt.values())
            note = ("sum of the six HIP kernels with precomputed mixing weights; excludes the "
                    "weight network, noise draws and autograd glue")
        # HBM bytes per launch counted by the PMC passes committed under profiles/
        # (rocprofv3 cannot run inside this process); null for shapes not profiled
        traffic_table = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")) as fh:
                traffic_table = json.load(fh).get(f"T{T}_B{B}_n{n}_p{p}_K{K}", {})
        except OSError:
            pass
        name, sec, flops, by = kal
        # which roof bounds the Kalman kernel: arithmetic intensity against the ridge
        # (157.3 TFLOP/s / 8 TB/s = 19.7 flop/B)
        valu_bound = flops / by > VALU_FP32_PEAK_TF * 1e3 / HBM_PEAK_GBS
        kal_roof = {
            "kernel": name, "bound": "valu" if valu_bound else "hbm",
            "achieved": flops / sec / 1e12 if valu_bound else by / sec / 1e9,
            "peak": VALU_FP32_PEAK_TF if valu_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if valu_bound else "GB/s",
            "traffic": traffic_table.get(name), "algorithmic_bytes_per_launch": by,
            "algorithmic_flops_per_launch": flops, "avg_launch_us": sec * 1e6,
            "valu_frac": flops / sec / 1e12 / VALU_FP32_PEAK_TF,
            "hbm_frac": by / sec / 1e9 / HBM_PEAK_GBS}
        kal_roof["frac"] = kal_roof["achieved"] / kal_roof["peak"]
        result["kalman_roofline"] = kal_roof
        result["roofline"] = dict(kal_roof)
        if not args.kernels_only:
            # the weight network's kernels: the longest of the whole step (bound by the bytes
            # they stream with split products, by the fp32 matrix core with exact ones)
            lk = time_lstm_kernels(ssm, a)
            if lk is not None:
                result["weight_network_kernels"] = lk
                top = max(lk, key=lambda k_: lk[k_]["us"])
                if lk[top]["us"] * 1e-6 > sec:
                    r = lk[top]
                    result["roofline"] = {
                        "kernel": r["kernel"], "bound": r["bound"], "achieved": r["achieved"],
                        "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                        "traffic": traffic_table.get(top), "avg_launch_us": r["us"],
                        "algorithmic_flops_per_launch": r["flops_per_launch"],
                        "algorithmic_bytes_per_launch": r["hbm_bytes_per_launch"],
                        "share_of_step": r["us"] * 1e-6 / step_s}
        result["kernels"] = kernels
        result["kernel_only"] = {
            "us_per_step": total_k * 1e6, "sequence_steps_per_s": units / total_k,
            "valu_frac": step_flops / total_k / 1e12 / VALU_FP32_PEAK_TF, "note": note}
    if world == 1 and not args.no_cpu_baseline and not args.kernels_only:
        result["cpu_baseline"] = cpu_baseline(T, B, args.net, args.burn_in, n, p, K)
    return result


if __name__ == "__main__":
    main()
