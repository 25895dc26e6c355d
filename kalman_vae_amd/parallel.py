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
    ``grad = None`` (the optimizer skips them, as on a single device) and their slices
    stay zero.  ``async_op=True`` returns the collective's work handle: the reduction runs
    on the backend's own stream, and the caller waits on the handle before reading the
    gradients (or before the next ``all_reduce``).
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        p0 = self.params[0]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=p0.dtype, device=p0.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._pending = None

    def wait(self):
        """Block the current stream until the last asynchronous all-reduce is done."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def pack(self):
        """Gather the gradients that do not already live in the buffer (one launch)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
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
            backend = dist.get_backend(group)
            if self.flat.is_cuda and backend == "gloo":
                # gloo reduces host memory (its device path hangs on this ROCm build);
                # only rehearsals and tests pair gloo with GPU tensors
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                if average:
                    host /= dist.get_world_size(group)
                self.flat.copy_(host)
            elif backend == "nccl":
                op = dist.ReduceOp.AVG if average else dist.ReduceOp.SUM
                work = dist.all_reduce(self.flat, op=op, group=group, async_op=async_op)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                if average:
                    self.flat.div_(dist.get_world_size(group))
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
