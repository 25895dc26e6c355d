"""World-size-2 CPU test (gloo) of the data-parallel path: each rank takes its
shard of the batch, computes gradients of the Kalman objective for the shard
(with the oracle as the compute here -- the package has no CPU kernels), and
the flat-bucket all-reduce must reproduce the full-batch gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kalman_vae_amd.parallel import FlatGradBucket, shard_batch, shard_bounds


def test_shard_bounds_cover_everything():
    for total in (1, 7, 64, 1000):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(total, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _objective(model, a, eps):
    from oracle import kalman_oracle as O
    model.weight_model.clear_hidden_state()
    obj, _ = O.kalman_objective(
        a, model._mat_A_K, model._mat_C_K, model.mat_Q, model.mat_R, model.initial_state_mean,
        model.initial_state_covariance, model.weight_model, eps,
        sequence_operation="mean", batch_operation="sum")
    return obj


def _make(seed=0):
    from kalman_vae_amd import StateSpaceModel
    torch.manual_seed(seed)
    model = StateSpaceModel(2, 4, 3, "lstm", hidden_dim=8).double()
    a = 0.7 * torch.randn(6, 10, 2, dtype=torch.float64)
    eps = torch.randn(6, 10, 4, dtype=torch.float64)
    return model, a, eps


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model, a, eps = _make()
    a_r, eps_r = shard_batch(a, world, rank), shard_batch(eps, world, rank)
    _objective(model, a_r, eps_r).backward()
    bucket = FlatGradBucket(model.parameters())
    flat = bucket.all_reduce(average=False)
    if rank == 0:
        torch.save(flat.clone(), out)
    dist.destroy_process_group()


def test_flat_bucket_allreduce_matches_full_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "flat.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    model, a, eps = _make()
    _objective(model, a, eps).backward()
    want = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in model.parameters() if p.requires_grad])
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 1e-10 * max(1.0, float(want.abs().max()))


def test_bucket_gathers_with_one_launch_and_copies_nothing_back(monkeypatch):
    """One fused multi-tensor copy per step, gradients re-pointed at the flat buffer (no
    copy back), parameters without a gradient left at grad = None."""
    model, a, eps = _make()
    _objective(model, a, eps).backward()
    had = {n for n, p in model.named_parameters() if p.grad is not None}
    assert "a_0" not in had and "_mat_Q_L" not in had      # no gradient on a single device
    want = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    bucket = FlatGradBucket(model.parameters())
    calls = {"foreach": 0, "copy": 0}
    real_foreach, real_copy = torch._foreach_copy_, torch.Tensor.copy_

    def count_foreach(dst, src, *a_, **k_):
        calls["foreach"] += 1
        return real_foreach(dst, src, *a_, **k_)

    def count_copy(self, *a_, **k_):
        calls["copy"] += 1
        return real_copy(self, *a_, **k_)

    monkeypatch.setattr(torch, "_foreach_copy_", count_foreach)
    monkeypatch.setattr(torch.Tensor, "copy_", count_copy)
    bucket.all_reduce(average=True)
    assert calls == {"foreach": 1, "copy": 0}
    by_name = dict(model.named_parameters())
    for (n, p), v in zip(((n, by_name[n]) for n in by_name if by_name[n].requires_grad),
                         bucket.views):
        if n in had:
            assert p.grad.data_ptr() == v.data_ptr()          # a view of the bucket
            assert torch.equal(p.grad, want[n])
        else:
            assert p.grad is None and float(v.abs().max()) == 0.0
    bucket.all_reduce(average=True)                            # gradients already in place
    assert calls == {"foreach": 1, "copy": 0}


# -- DataParallelStep over the full model's parameters ------------------------------
def _full_model(seed=3):
    from kalman_vae_amd import KalmanVariationalAutoencoder
    torch.manual_seed(seed)
    model = KalmanVariationalAutoencoder((16, 16), 1, 2, 4, 3, 0.9, 0.9, True, 0.8, "mlp",
                                         decoder_type="bernoulli").double()
    xs = (torch.rand(5, 4, 1, 16, 16, dtype=torch.float64) > 0.5).double()
    eps = torch.randn(5, 4, 4, dtype=torch.float64)
    return model, xs, eps


def _full_objective(model, xs, eps):
    """Encoder mean -> Kalman objective (oracle: the package computes it on the GPU
    only) + decoder log-likelihood, batch mean: the same composition as the model's
    elbo, in plain PyTorch."""
    from oracle import kalman_oracle as O
    T, B = xs.shape[:2]
    a = model.encoder(xs.reshape(T * B, *xs.shape[2:])).loc.view(T, B, -1)
    ssm = model.state_space_model
    kal, _ = O.kalman_objective(a, ssm._mat_A_K, ssm._mat_C_K, ssm.mat_Q, ssm.mat_R,
                                ssm.initial_state_mean, ssm.initial_state_covariance,
                                ssm.weight_model, eps, sequence_operation="mean",
                                batch_operation="mean")
    rec = model.decoder(a.view(T * B, -1)).log_prob(xs.reshape(T * B, *xs.shape[2:]))
    return kal + 0.3 * rec.view(T, B, -1).sum(-1).mean()


def _step_worker(rank, world, port, out):
    from kalman_vae_amd.parallel import DataParallelStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model, xs, eps = _full_model()
    xs_r, eps_r = shard_batch(xs, world, rank), shard_batch(eps, world, rank)
    step = DataParallelStep(model, torch.optim.SGD(model.parameters(), lr=0.05))
    for _ in range(2):
        step(lambda: _full_objective(model, xs_r, eps_r))
    if rank == 1:       # any rank: all of them must hold the same parameters
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    dist.destroy_process_group()


def test_data_parallel_step_matches_single_process(tmp_path):
    from kalman_vae_amd.parallel import DataParallelStep
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "params.pt")
    mp.spawn(_step_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    model, xs, eps = _full_model()
    step = DataParallelStep(model, torch.optim.SGD(model.parameters(), lr=0.05))
    first = None
    for _ in range(2):
        obj = step(lambda: _full_objective(model, xs, eps))
        first = obj if first is None else first
    assert float(obj) > float(first)            # the step ascends the objective
    for k, v in model.state_dict().items():
        assert float((got[k] - v).abs().max()) < 1e-10 * max(1.0, float(v.abs().max())), k


# -- masked imputation: the batch is sharded, no collective on the data path --------
def _masked_problem(seed=5):
    from kalman_vae_amd import StateSpaceModel
    torch.manual_seed(seed)
    model = StateSpaceModel(2, 4, 3, "mlp", hidden_dim=12).double().eval()
    T, B = 7, 11                                   # ragged: ranks get 6 and 5 sequences
    a = 0.7 * torch.randn(T, B, 2, dtype=torch.float64)
    mask = (torch.rand(T, B) < 0.6).double()
    mask[0] = 1.0
    return model, a, mask


def _masked_filter_host(model, a, mask):
    """The device source of the one-launch masked filter (perceptron network inside the
    recursion), compiled for the host by tests/hostsim: the CPU stand-in for the kernel."""
    import ctypes as C
    from kalman_vae_amd import _capi as capi
    from kalman_vae_amd.functions import pack_mlp2
    from kernel_harness import hostsim_lib
    T, B, p = a.shape
    n, K = model.z_dim, model.K
    new = lambda *s: torch.zeros(s, dtype=a.dtype)
    out = dict(m=new(T, B, n), P=new(T, B, n, n), mn=new(T, B, n), Pn=new(T, B, n, n),
               As=new(T + 1, B, n, n), Cs=new(T + 1, B, p, n), a_w=new(T, B, p),
               alpha=new(T + 1, B, K))
    out["alpha"][0] = 1.0 / K
    status = torch.zeros(1, dtype=torch.int32)
    with torch.no_grad():
        keep = [t.detach().contiguous() for t in (
            a, mask, model.mat_A_K, model.mat_C_K, model.mat_Q, model.mat_R,
            model.initial_state_mean, model.initial_state_covariance,
            pack_mlp2(model.weight_model.layers[0], model.weight_model.layers[1]))]
    io = capi.fill(capi.FilterFwd(), a=keep[0], alpha=out["alpha"], mask=keep[1], A_K=keep[2],
                   C_K=keep[3], Q=keep[4], R=keep[5], m_in=keep[6], P_in=keep[7],
                   carry_per_sequence=0, mean_filt=out["m"], cov_filt=out["P"],
                   mean_pred=out["mn"], cov_pred=out["Pn"], mat_As=out["As"], mat_Cs=out["Cs"],
                   status=status)
    net = capi.fill(capi.MaskedMlp(), hidden=model.weight_model.layers[0].out_features,
                    packed=keep[8], eps_z=None, L_Q=None, eps_a=None, L_R=None, a_w=out["a_w"])
    fn = hostsim_lib().hostsim_filter_forward_masked_mlp
    fn.argtypes = [C.POINTER(capi.Dims), C.POINTER(capi.FilterFwd), C.POINTER(capi.MaskedMlp)]
    fn.restype = C.c_int
    dims = capi.Dims(T, B, n, p, K, capi.KVAE_F64, 1, 0)
    assert fn(C.byref(dims), C.byref(io), C.byref(net)) == 0 and int(status[0]) == 0
    return torch.cat([out[k].reshape(T, B, -1) for k in ("m", "P", "mn", "Pn", "a_w")], dim=2)


def _impute_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model, a, mask = _masked_problem()
    mine = _masked_filter_host(model, shard_batch(a, world, rank).contiguous(),
                               shard_batch(mask, world, rank).contiguous())
    torch.save(mine, f"{out}.{rank}")          # every rank keeps its shard: nothing is exchanged
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_masked_imputation_matches_full_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "shard")
    mp.spawn(_impute_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.cat([torch.load(f"{out}.{r}") for r in range(2)], dim=1)
    model, a, mask = _masked_problem()
    want = _masked_filter_host(model, a, mask)
    assert got.shape == want.shape
    assert torch.equal(got, want)              # sequences are independent: bit-identical


# -- two steps, with a parameter that stops receiving a gradient --------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_step_worker(rank, world, port, out, skip_on):
    """Step 1: both parameters get a gradient on every rank.  Step 2: `w2` gets none on the
    ranks in `skip_on`."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    torch.manual_seed(0)
    w1 = torch.nn.Parameter(torch.randn(3, dtype=torch.float64))
    w2 = torch.nn.Parameter(torch.randn(2, dtype=torch.float64))
    bucket = FlatGradBucket([w1, w2])
    x = torch.arange(1.0, 4.0, dtype=torch.float64) * (rank + 1)
    ((w1 * x).sum() + (w2 * (rank + 2.0)).sum()).backward()
    bucket.all_reduce(average=True)
    step1 = (w1.grad.clone(), w2.grad.clone())
    w1.grad, w2.grad = None, None                      # zero_grad(set_to_none=True)
    loss = (w1 * x * x).sum()
    if rank not in skip_on:
        loss = loss + (w2 * 5.0).sum()
    loss.backward()
    error = None
    try:
        bucket.all_reduce(average=True)
    except RuntimeError as err:
        error = str(err)
    torch.save({"step1": step1, "w1": w1.grad, "w2": w2.grad, "w2_slice": bucket.views[1].clone(),
                "error": error}, f"{out}.{rank}")
    dist.destroy_process_group()


def test_parameter_without_gradient_on_every_rank_leaves_a_zero_slice(tmp_path):
    """A parameter that had a gradient in step 1 and has none in step 2 (on all ranks): its
    slice of the bucket must not carry step 1's average into step 2's reduction."""
    out = str(tmp_path / "two")
    mp.spawn(_two_step_worker, args=(2, _free_port(), out, (0, 1)), nprocs=2, join=True)
    for rank in range(2):
        got = torch.load(f"{out}.{rank}")
        assert got["error"] is None
        assert torch.equal(got["step1"][0], torch.tensor([1.5, 3.0, 4.5], dtype=torch.float64))
        assert torch.equal(got["step1"][1], torch.full((2,), 2.5, dtype=torch.float64))
        assert torch.equal(got["w1"], torch.tensor([2.5, 10.0, 22.5], dtype=torch.float64))
        assert got["w2"] is None and float(got["w2_slice"].abs().max()) == 0.0


def test_ranks_that_disagree_on_a_gradient_are_told_so(tmp_path):
    """Rank 1 produces no gradient for a parameter that rank 0 steps on: both ranks raise
    instead of letting the replicas drift apart."""
    out = str(tmp_path / "two")
    mp.spawn(_two_step_worker, args=(2, _free_port(), out, (1,)), nprocs=2, join=True)
    for rank in range(2):
        got = torch.load(f"{out}.{rank}")
        assert got["error"] is not None and "disagree" in got["error"] and "[1]" in got["error"]


def test_async_reduction_is_waited_for_only_before_the_bucket_is_refilled(monkeypatch):
    """The order of an overlapped step: the all-reduce of step k is launched and NOT waited
    for (the next replay of the training step runs beside it); the next all_reduce() call
    waits for it before it packs step k+1's gradients into the same buffer; the flags of a
    finished reduction are looked at, without blocking the launch path, one call later."""
    import kalman_vae_amd.parallel as par
    log = []

    class Work:
        def __init__(self, k):
            self.k = k

        def wait(self):
            log.append(f"wait{self.k}")

    count = {"k": 0}

    def fake_all_reduce(tensor, op=None, group=None, async_op=False):
        count["k"] += 1
        log.append(f"launch{count['k']}")
        assert async_op
        return Work(count["k"])

    monkeypatch.setattr(par.dist, "is_available", lambda: True)
    monkeypatch.setattr(par.dist, "is_initialized", lambda: True)
    monkeypatch.setattr(par.dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(par.dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(par.dist, "all_reduce", fake_all_reduce)
    real_foreach = torch._foreach_copy_
    monkeypatch.setattr(torch, "_foreach_copy_",
                        lambda d, s, *a, **k: (log.append("pack"), real_foreach(d, s, *a, **k))[1])
    w = torch.nn.Parameter(torch.ones(3))
    bucket = par.FlatGradBucket([w])
    for step in range(1, 4):
        log.append(f"replay{step}")
        w.grad = torch.full((3,), float(step))
        handle = bucket.all_reduce(average=True, async_op=True)
        assert isinstance(handle, Work) and torch.equal(w.grad, torch.full((3,), float(step)))
    bucket.wait()
    bucket.check()
    assert log == ["replay1", "pack", "launch1",
                   "replay2", "wait1", "pack", "launch2",
                   "replay3", "wait2", "pack", "launch3", "wait3"]
    # a flag that comes back as "one of two ranks": copied to the host when the reduction is
    # waited for, reported by check() or by the wait after that
    w.grad = torch.full((3,), 4.0)
    bucket.all_reduce(average=True, async_op=True)
    bucket.flags.fill_(0.5)
    bucket.all_reduce(average=True, async_op=True)        # waits for the poisoned one
    with pytest.raises(RuntimeError, match="disagree"):
        bucket.check()
    bucket.flags.fill_(0.5)
    bucket.all_reduce(average=True, async_op=True)
    with pytest.raises(RuntimeError, match="disagree"):
        bucket.all_reduce(average=True, async_op=True)
