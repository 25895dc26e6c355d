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

    The Kalman model has ~0.2-0.25 M parameters (< 1 MB in fp32): at that size
    a collective is latency-bound, so a single bucket (one launch, one ring or
    tree pass) beats per-tensor or multi-bucket reduction.  Gradients are
    copied in with one fused multi-tensor copy and copied back the same way.
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

    def pack(self):
        self.flat.zero_()
        src = [p.grad for p in self.params if p.grad is not None]
        dst = [v for p, v in zip(self.params, self.views) if p.grad is not None]
        if src:
            torch._foreach_copy_(dst, src)

    def unpack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)

    def all_reduce(self, average=True, group=None):
        """Sum (or average) the gradients over all ranks, in place."""
        self.pack()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if self.flat.is_cuda and dist.get_backend(group) == "gloo":
                # gloo reduces host memory (its device path hangs on this ROCm build);
                # only rehearsals and tests pair gloo with GPU tensors
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))
        self.unpack()
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
