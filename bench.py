#!/usr/bin/env python3
"""Benchmark of the Kalman filter + RTS smoother + Kalman-ELBO hot path,
forward and backward, on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one forward+backward pass of the path over one batch of synthetic
bouncing-ball-shaped sequences: weight network -> mixing weights -> filter ->
smoother -> ELBO terms -> aggregate -> backward through all of it (plus, for
N > 1, one flat-bucket RCCL all-reduce of the parameter gradients).  Inputs are
resident in HBM before the timed region.  Per-GPU work is fixed as N grows
(weak scaling): every rank processes its own B sequences.

Rank 0 prints ONE JSON line.  `value` = sequence-steps/s of the whole job
(N * T * B * steps / max-over-ranks time).  Additional objects:
  roofline      dominant kernel: algorithmic bytes / measured duration vs HBM peak
  cpu_baseline  the oracle (a per-step PyTorch restatement of the reference
                algorithm) timed on this machine's host cores on the same workload
  kernels       per-kernel average durations (HIP events) and rates
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


def time_lstm_kernels(ssm, a, iters=20):
    """HIP-event durations of the two persistent LSTM kernels of the weight network
    (the step's longest kernels) at the bench shape, launched through the C ABI on
    the current stream; None when the network is not the default two-layer H=128 LSTM."""
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
    w = lambda nme: getattr(lstm, nme).detach()
    whh0t, wih1t, whh1t, b0, b1 = LF._lstm2_operands(
        w("weight_hh_l0"), w("bias_ih_l0"), w("bias_hh_l0"), w("weight_ih_l1"), w("weight_hh_l1"),
        w("bias_ih_l1"), w("bias_hh_l1"))
    new = lambda *s: torch.empty(*s, dtype=x.dtype, device=dev)
    g0, h0, c0, g1, h1, c1 = (new(T, B, 4 * H), new(T, B, H), new(T, B, H), new(T, B, 4 * H),
                              new(T, B, H), new(T, B, H))
    dG0, dG1, dx = torch.empty_like(g0), torch.empty_like(g1), torch.empty_like(x)
    gh = torch.randn(T, B, H, dtype=x.dtype, device=dev) * 1e-3
    part = new(lib.kvae_lstm2_backward_blocks(B), 4 * H * (4 + P))
    wih0 = w("weight_ih_l0").contiguous()
    st = capi.current_stream_ptr()

    def fwd():
        capi.check(lib.kvae_lstm2_forward(
            T, B, P, x.data_ptr(), wih0.data_ptr(), b0.data_ptr(), whh0t.data_ptr(),
            wih1t.data_ptr(), whh1t.data_ptr(), b1.data_ptr(), g0.data_ptr(), h0.data_ptr(),
            c0.data_ptr(), g1.data_ptr(), h1.data_ptr(), c1.data_ptr(), None, None, st))

    def bwd():
        capi.check(lib.kvae_lstm2_backward(
            T, B, P, gh.data_ptr(), x.data_ptr(), g0.data_ptr(), c0.data_ptr(), g1.data_ptr(),
            c1.data_ptr(), wih0.data_ptr(), w("weight_hh_l0").contiguous().data_ptr(),
            w("weight_ih_l1").contiguous().data_ptr(), w("weight_hh_l1").contiguous().data_ptr(),
            dG0.data_ptr(), dG1.data_ptr(), dx.data_ptr(), part.data_ptr(), st))

    flops = 2.0 * T * B * 4 * H * 3 * H     # one recurrent product per layer + layer 1's input product
    out = {}
    for name, fn in (("lstm_forward", fwd), ("lstm_backward", bwd)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) / iters * 1e-3
        out[name] = {"us": sec * 1e6, "bound": "mfma", "achieved": flops / sec / 1e12,
                     "peak": MFMA_FP32_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / sec / 1e12 / MFMA_FP32_PEAK_TF, "flops_per_launch": flops}
    return out


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


def cpu_baseline(T, B, net, burn_in, n=4, p=2, K=3, budget_s=20.0):
    """The oracle (per-step PyTorch ops, autograd backward) on the host cores."""
    from kalman_vae_amd import StateSpaceModel
    from oracle import kalman_oracle as O
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
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 30):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": T * B / med, "unit": "sequence-steps/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{len(times)} fwd+bwd iterations of T={T} B={B} (median {med*1e3:.1f} ms), "
                      f"oracle/kalman_oracle.py with the {net} weight network, torch "
                      f"{torch.__version__} CPU, {os.cpu_count()} logical CPUs visible"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--net", default="lstm", choices=["lstm", "mlp"])
    ap.add_argument("--burn-in", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--state-dim", type=int, default=4)
    ap.add_argument("--obs-dim", type=int, default=2)
    ap.add_argument("--bases", type=int, default=3)
    ap.add_argument("--kernels-only", action="store_true",
                    help="time only the six HIP kernels with synthetic mixing weights "
                         "(softmax of N(0,1)); no PyTorch weight network, no autograd glue")
    ap.add_argument("--kernel-iters", type=int, default=30)
    ap.add_argument("--separate-kernels", action="store_true",
                    help="run filter, smoother and ELBO terms as six separate kernels with all "
                         "their tensors materialised (the drop-in API's path) instead of the "
                         "fused objective")
    ap.add_argument("--no-hip-graph", action="store_true",
                    help="run the step eagerly instead of replaying it as one HIP graph")
    args = ap.parse_args()

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

    from kalman_vae_amd import SampleControl, kalman_elbo
    from kalman_vae_amd.parallel import FlatGradBucket

    T, B = args.seq_len, args.batch
    dtype = torch.float32
    torch.manual_seed(0)   # same initial parameters on every rank
    n, p, K = args.state_dim, args.obs_dim, args.bases
    ssm = build_model(dev, dtype, args.net, n, p, K, fused=not args.separate_kernels)
    a = bouncing_ball_observations(T, B, p, dev, seed=1000 + rank).to(dtype).requires_grad_(True)
    bucket = FlatGradBucket(ssm.parameters()) if world > 1 else None
    sc = SampleControl()

    def eager_step():
        ssm.zero_grad(set_to_none=True)
        a.grad = None
        obj, _ = kalman_elbo(ssm, a, sc, burn_in=args.burn_in)
        (-obj).backward()
        if bucket is not None:
            bucket.all_reduce(average=True)

    # The whole step (weight network, filter, smoother, ELBO terms and their
    # backward: ~90 launches) replayed as one HIP graph; same kernels, same work.
    graphed, graph_note = None, None
    if not args.no_hip_graph and not args.kernels_only:
        from kalman_vae_amd.graphed import GraphedElboStep
        try:
            graphed = GraphedElboStep(ssm, a, sc, burn_in=args.burn_in)
        except Exception as err:   # capture refused or failed: time the eager step instead
            graph_note = f"{type(err).__name__}: {err}"
            torch.cuda.synchronize()

    def graph_step():
        graphed(None)            # the synthetic batch already is the static input
        if bucket is not None:
            bucket.all_reduce(average=True)

    step = graph_step if graphed is not None else eager_step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.kernels_only:
        kt = time_kernels(ssm, a, args.burn_in, iters=args.kernel_iters, synthetic_alpha=True)
        elapsed = sum(kt.values()) * args.steps
    else:
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    result = None
    if rank == 0:
        units = T * B
        ms_per_step = elapsed / args.steps * 1e3
        value = world * units * args.steps / elapsed
        result = {
            "metric": "filter+smoother sequence-steps/sec (fwd+bwd)",
            "value": value, "unit": "sequence-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"bouncing-ball shapes T={T} a_dim={p} z_dim={n} K={K}, "
                                   f"{B} sequences per GPU, "
                                   + ("synthetic mixing weights, HIP kernels only, "
                                      if args.kernels_only else f"{args.net} weight network, ")
                                   + "filter+RTS smoother+ELBO terms fwd+bwd",
                       "seq_len": T, "batch_per_gpu": B, "global_batch": B * world,
                       "burn_in": args.burn_in,
                       "parallelism": f"dp{world}" + (" (rehearsal: one GPU, gloo)" if rehearsal else ""),
                       "hip_graph": graphed is not None},
        }
        if graph_note:
            result["config"]["hip_graph_note"] = graph_note
        if not args.no_kernel_timing:
            if not args.kernels_only:
                kt = time_kernels(ssm, a, args.burn_in, iters=args.kernel_iters)
            fl = flops_forward(n, p, K)
            kb = kernel_bytes(n, p, K, 4)
            kflops = {"filter_forward": fl["filter"], "filter_backward": 2 * fl["filter"],
                      "smooth_forward": fl["smooth"], "smooth_backward": 2 * fl["smooth"],
                      "elbo_forward": fl["elbo"], "elbo_backward": 2 * fl["elbo"]}
            kernels = {}
            for name, sec in kt.items():
                kernels[name] = {"us": sec * 1e6, "GBps": kb[name] * units / sec / 1e9,
                                 "TFLOPs": kflops[name] * units / sec / 1e12}
            dom = max(kt, key=kt.get)
            achieved = kb[dom] * units / kt[dom] / 1e9
            # HBM bytes per launch counted by the PMC passes committed under profiles/
            # (rocprofv3 cannot run inside this process); null for shapes not profiled
            traffic = None
            try:
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                       "pmc_hbm_traffic.json")) as fh:
                    traffic = json.load(fh).get(f"T{T}_B{B}_n{n}_p{p}_K{K}", {}).get(dom)
            except OSError:
                pass
            result["roofline"] = {
                "kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": kb[dom] * units,
                "avg_launch_us": kt[dom] * 1e6,
                "valu_frac": kflops[dom] * units / kt[dom] / 1e12 / VALU_FP32_PEAK_TF,
            }
            total_k = sum(kt.values())
            if not args.kernels_only:
                # the weight network's recurrence: the longest kernels of the whole step,
                # bound by the fp32 matrix core rather than by HBM
                lk = time_lstm_kernels(ssm, a)
                if lk is not None:
                    result["weight_network_kernels"] = lk
            result["kernels"] = kernels
            result["kernel_only"] = {
                "us_per_step": total_k * 1e6,
                "sequence_steps_per_s": units / total_k,
                "valu_frac": 3 * sum(fl.values()) * units / total_k / 1e12 / VALU_FP32_PEAK_TF,
                "note": "sum of the six HIP kernels with precomputed mixing weights; excludes "
                        "the PyTorch weight network, noise draws and autograd glue",
            }
        if world == 1 and not args.no_cpu_baseline and not args.kernels_only:
            result["cpu_baseline"] = cpu_baseline(T, B, args.net, args.burn_in, n, p, K)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
