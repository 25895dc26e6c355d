"""One training-style evaluation of the Kalman ELBO (forward + backward) as a
HIP graph.

At the reference's shapes (T=20, a few thousand sequences) a step is some 90
short kernels and the Python/autograd dispatch of those launches costs more
than the kernels run; replaying the captured graph removes that cost.  Capture
needs static shapes and no host synchronisation, so the module's
``check_numerics`` is switched off during capture: call
``kalman_vae_amd.read_status(device)`` when you want to look at the failure
bits (one sync).
"""
import torch

from . import _capi as capi
from .elbo import kalman_elbo


class GraphedElboStep:
    """Captures ``objective = kalman_elbo(ssm, as_, ...)`` and
    ``(-objective).backward()`` for a fixed ``as_`` shape.

    ``step(as_)`` copies ``as_`` into the static input, replays the graph and
    returns the objective (a tensor that the next replay overwrites; ``step.info`` is the
    ``info`` dictionary of the captured call, whose tensors are overwritten alike).  After a
    replay the gradients are in ``p.grad`` of the module's parameters and in
    ``step.as_grad``; both are overwritten, not accumulated, by every replay
    (``p.grad`` is re-pointed at the graph's static tensors after each replay).
    Random draws inside the step advance with every replay (torch registers the
    generator with the graph).
    """

    def __init__(self, ssm, as_example, sample_control, warmup=3, **elbo_kwargs):
        if not as_example.is_cuda:
            raise RuntimeError("GraphedElboStep needs a GPU tensor (there is no CPU path)")
        self.ssm = ssm
        self.static_as = as_example.detach().clone().requires_grad_(True)
        self._kwargs = dict(elbo_kwargs)
        self._sc = sample_control
        self._checked = ssm.check_numerics
        self._minus_one = None
        ssm.check_numerics = False
        try:
            side = torch.cuda.Stream(device=as_example.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):           # library workspaces, lazy inits
                    self._eager()
            torch.cuda.current_stream().wait_stream(side)
            ssm.zero_grad(set_to_none=True)
            self.static_as.grad = None
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            # thread_local: other threads (e.g. the RCCL watchdog of torch.distributed)
            # may keep making runtime calls while this thread captures
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                self.objective = self._eager()
        finally:
            ssm.check_numerics = self._checked
        lib = capi.load()
        raw = self.graph.raw_cuda_graph()
        self.kernel_nodes = lib.kvae_graph_count_kernel_nodes(raw)
        memsets = lib.kvae_graph_count_memset_nodes(raw)
        if memsets != 0:
            # A memset node is not reliably ordered before the kernel that depends on
            # it when a graph is replayed on ROCm 7.0 (seen with library reductions
            # that zero a semaphore: the result is silently not written).
            self.graph = None
            raise RuntimeError(
                f"the captured step contains {memsets} memset node(s) (a torch reduction "
                "over many rows, e.g. the bias gradient of a torch.nn.Linear); their "
                "replay order is unreliable on this ROCm version, so the step is not "
                "run as a HIP graph - use the eager step for this configuration")
        self.graph.instantiate()
        self.as_grad = self.static_as.grad
        # the graph writes its gradients into these tensors on every replay
        self._static_grads = [(p, p.grad) for p in ssm.parameters() if p.grad is not None]

    def _eager(self):
        self.ssm.zero_grad(set_to_none=True)
        self.static_as.grad = None
        # (info: static tensors too, e.g. info["elbo_terms"]; every replay overwrites them)
        objective, self.info = kalman_elbo(self.ssm, self.static_as, self._sc, **self._kwargs)
        # gradients of the loss -objective: a constant -1 as the upstream gradient
        # instead of a negation kernel, its adjoint and the seed fill
        if self._minus_one is None:
            self._minus_one = torch.full((), -1.0, dtype=objective.dtype, device=objective.device)
        objective.backward(self._minus_one)
        return objective.detach()

    def __call__(self, as_=None):
        if as_ is not None:
            with torch.no_grad():
                self.static_as.copy_(as_)
        self.graph.replay()
        for p, g in self._static_grads:      # survive zero_grad(set_to_none=True) by the caller
            p.grad = g
        return self.objective
