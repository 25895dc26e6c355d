"""The masked training step with the default LSTM network (ELBO forward + backward, T = 20)
captured as a HIP graph: replay against eager, objective and gradients, and both timed.
python tools/exp/graph_masked_lstm.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kalman_vae_amd import SampleControl, StateSpaceModel  # noqa: E402
from kalman_vae_amd.elbo import kalman_elbo  # noqa: E402
from kalman_vae_amd.graphed import GraphedElboStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
ssm = StateSpaceModel(2, 4, 3, "lstm").to(dev).train()
a = torch.randn(20, B, 2, device=dev)
mask = (torch.rand(20, B, device=dev) < 0.7).float()
sc = SampleControl()


def eager():
    ssm.zero_grad(set_to_none=True)
    x = a.clone().requires_grad_(True)
    obj, _ = kalman_elbo(ssm, x, sc, observation_mask=mask)
    (-obj).backward()
    return obj.detach().clone(), x.grad.clone(), [p.grad.clone() for p in ssm.parameters() if p.grad is not None]


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t_eager = timed(eager)
step = GraphedElboStep(ssm, a, sc, observation_mask=mask)
t_graph = timed(lambda: step(a))
print(f"B={B}: masked ELBO step, LSTM network: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms "
      f"({step.kernel_nodes} kernel nodes)")
