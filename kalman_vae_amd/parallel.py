"""Data-parallel helpers: sequences are independent, so a batch is split along
B across ranks (one process per GPU) with no collective on the data path; the
only exchange is one all-reduce of a flat gradient bucket per step
(``torch.distributed``: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on
CPU for tests).  The reference itself is single-device (examples/train.py
runs on one ``--device``); this module has no counterpart there.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous, balanced [begin, end) slice of `total` items for `rank`;
    the first total % world_size ranks get one extra item."""
    base, extra = divmod(total, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(tensor, world_size, rank, dim=1):
    """Slice of a sequence-first tensor (T,B,...) along the batch axis."""
    begin, end = shard_bounds(tensor.shape[dim], world_size, rank)
    return tensor.narrow(dim, begin, end - begin)


class FlatGradBucket:
    """All gradients of `params` in ONE flat buffer and one all-reduce.

    The Kalman model has ~0.2-0.25 M parameters (< 1 MB in fp32): at that size a
    collective is latency-bound, so a single bucket (one launch, one ring or tree pass)
    beats per-tensor or multi-bucket reduction.

    Per step: ONE fused multi-tensor copy gathers the gradients autograd produced into
    the flat buffer, the all-reduce averages it in place (``ReduceOp.AVG`` where the
    backend has it: no separate division), and every parameter's ``.grad`` is re-pointed
    at its slice of the buffer -- no copy back.  Parameters that received no gradient keep
    ``grad = None`` (the optimizer skips them, as on a single device); their slices are
    zeroed before the reduction, whatever an earlier step left there.

    All ranks must produce gradients for the SAME parameters in a step: a rank without a
    gradient for a parameter would keep ``grad = None`` while the others step on it, and the
    replicas would drift apart.  One flag per parameter rides at the end of the buffer
    through the same all-reduce; a flag that comes back neither 0 nor "all ranks" raises
    ``RuntimeError`` -- at once for a blocking reduction; for an asynchronous one its flags
    are copied to the host without blocking when it is waited for and looked at one
    ``all_reduce()`` later (or by ``check()``), so that the host never stalls on the reduction
    it overlaps.

    ``async_op=True`` returns the collective's work handle: the reduction runs
    on the backend's own stream, and the caller waits on the handle before reading the
    gradients (or before the next ``all_reduce``).
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        p0 = self.params[0]
        self.numel = sum(p.numel() for p in self.params)
        self.storage = torch.zeros(self.numel + len(self.params), dtype=p0.dtype, device=p0.device)
        self.flat = self.storage[:self.numel]           # the gradients
        self.flags = self.storage[self.numel:]          # 1 where this rank has a gradient
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._pending = None
        self._pattern = None            # which parameters had a gradient at the last pack()
        self._flag_src = None
        self._flags_stale = True        # the buffer's flags are not this rank's pattern
        self._clean = [True] * len(self.params)   # slice known to be all zero
        self._host_flags = None
        self._inflight = None           # (pattern, scale, world) of the reduction in flight
        self._copied = None             # (event, pattern, scale, world): its flags, on their way

    def wait(self):
        """Block the current stream until the last asynchronous all-reduce is done.  Its flags
        are copied to the host behind it without blocking; they are looked at by the wait()
        after this one, by which time the copy finished a step ago (looking at them at once
        would stall the host on the reduction it is meant to overlap), or by check()."""
        if self._pending is None:
            return
        self._pending.wait()
        self._pending = None
        self.check()                      # the reduction before this one
        if self._inflight is not None:
            if self._host_flags is None:
                self._host_flags = torch.empty(len(self.params), dtype=self.flags.dtype,
                                               pin_memory=self.flags.is_cuda)
            self._host_flags.copy_(self.flags, non_blocking=True)
            event = None
            if self.flags.is_cuda:
                event = torch.cuda.Event()
                event.record()
            self._copied = (event,) + self._inflight
            self._inflight = None

    def check(self):
        """Raise if the ranks of the last reduction that wait() has been called for disagreed
        on which parameters received a gradient (blocks until its flags are on the host)."""
        if self._copied is None:
            return
        event, pattern, scale, world = self._copied
        self._copied = None
        if event is not None:
            event.synchronize()
        self._verify(self._host_flags.tolist(), pattern, scale, world)

    @staticmethod
    def _verify(got, pattern, scale, world):
        bad = [i for i, (g, had) in enumerate(zip(got, pattern))
               if abs(g - (scale if had else 0.0)) > 0.25 * scale / world]
        if bad:
            raise RuntimeError(
                "data-parallel ranks disagree on which parameters received a gradient "
                f"(parameter indices {bad[:8]}{' ...' if len(bad) > 8 else ''}): "
                "every rank must produce gradients for the same parameters in a step")

    def pack(self):
        """Gather the gradients that do not already live in the buffer and this rank's flags
        (one launch); zero the slices of parameters without a gradient."""
        pattern = tuple(p.grad is not None for p in self.params)
        src, dst, stale = [], [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad
            if g is None:
                if not self._clean[i]:
                    stale.append(v)
                    self._clean[i] = True
            else:
                self._clean[i] = False
                if g.data_ptr() != v.data_ptr():
                    src.append(g)
                    dst.append(v)
        if pattern != self._pattern:
            self._flag_src = torch.tensor([float(f) for f in pattern], dtype=self.flags.dtype,
                                          device=self.flags.device)
            self._pattern = pattern
            self._flags_stale = True
        if self._flags_stale:
            src.append(self._flag_src)
            dst.append(self.flags)
            self._flags_stale = False
        if stale:
            torch._foreach_zero_(stale)
        if src:
            torch._foreach_copy_(dst, src)

    def bind(self):
        """Point the gradient of every parameter that has one at its slice of the buffer."""
        for p, v in zip(self.params, self.views):
            if p.grad is not None:
                p.grad = v

    def all_reduce(self, average=True, group=None, async_op=False):
        """Sum (or average) the gradients over all ranks, in place in the flat buffer."""
        self.wait()
        self.pack()
        work = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world = dist.get_world_size(group)
            backend = dist.get_backend(group)
            blocking = True
            if self.storage.is_cuda and backend == "gloo":
                # gloo reduces host memory (its device path hangs on this ROCm build);
                # only rehearsals and tests pair gloo with GPU tensors
                host = self.storage.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                if average:
                    host /= world
                self.storage.copy_(host)
            elif backend == "nccl":
                op = dist.ReduceOp.AVG if average else dist.ReduceOp.SUM
                work = dist.all_reduce(self.storage, op=op, group=group, async_op=async_op)
                blocking = not (async_op and work is not None)
            else:
                dist.all_reduce(self.storage, op=dist.ReduceOp.SUM, group=group)
                if average:
                    self.storage.div_(world)
            scale = 1.0 if average else float(world)
            self._flags_stale = not average       # a sum leaves "ranks with a gradient" there
            # slices no rank wrote stay zero; after a disagreement (an error) nothing is known
            if blocking:
                self._verify(self.flags.tolist(), self._pattern, scale, world)
            else:
                self._inflight = (self._pattern, scale, world)
        self.bind()
        if async_op and work is not None:
            self._pending = work
            return work
        return self.flat


class DataParallelStep:
    """One data-parallel training step of a model whose loss is ``-elbo``:

        step = DataParallelStep(model, optimizer)
        objective = step(lambda: model.elbo(sample_control, xs=local_frames, ...)[0])

    forward, backward, ONE all-reduce of all gradients (``FlatGradBucket``),
    optimizer step.  Every rank holds its own slice of the batch
    (``shard_batch``); the objective is a mean over the local batch, so
    averaging the gradients over ranks gives the gradient of the global-batch
    mean when the slices have equal size.  The reference trains on a single
    device (examples/train.py) and has no counterpart of this class.
    """

    def __init__(self, model, optimizer, group=None):
        self.model, self.optimizer, self.group = model, optimizer, group
        self.bucket = FlatGradBucket(model.parameters())

    def __call__(self, objective_fn):
        self.optimizer.zero_grad(set_to_none=True)
        objective = objective_fn()
        (-objective).backward()
        self.bucket.all_reduce(average=True, group=self.group)
        self.optimizer.step()
        return objective.detach()
