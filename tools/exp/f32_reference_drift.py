"""How far the float64 oracle's own arithmetic, run in float32 on the CPU, drifts from its
float64 result on a masked-gradient case of the randomised sweep (same construction as
tests/test_gpu_masked_grad.py::run_case): tells a float32 rounding miss of the kernels from a
defect.  python tools/exp/f32_reference_drift.py n p K T B H seed sample burn_in"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import rel_err  # noqa: E402
from oracle import kalman_oracle as O  # noqa: E402
from test_gpu_masked_grad import NAMES  # noqa: E402

n, p, K, T, B, H, seed = (int(x) for x in sys.argv[1:8])
sample, burn_in = sys.argv[8] == "1", int(sys.argv[9])
torch.manual_seed(seed + n * 13 + p)
dtype = torch.float32
first, last = torch.nn.Linear(p, H).to(dtype), torch.nn.Linear(H, K).to(dtype)
A_K = (0.9 * torch.eye(n) + 0.3 / n * torch.randn(K, n, n)).to(dtype)
C_K = (0.5 * torch.randn(K, p, n)).to(dtype)
Lq = 0.4 * torch.eye(n) + 0.05 * torch.randn(n, n)
Lr = 0.4 * torch.eye(p) + 0.05 * torch.randn(p, p)
Q, R = (Lq @ Lq.T + 0.01 * torch.eye(n)).to(dtype), (Lr @ Lr.T + 0.01 * torch.eye(p)).to(dtype)
Q, R = 0.5 * (Q + Q.T), 0.5 * (R + R.T)
m0, P0 = (0.3 * torch.randn(n)).to(dtype), (1.5 * torch.eye(n)).to(dtype)
a = torch.randn(T, B, p).to(dtype)
mask = (torch.rand(T, B) < 0.6).to(dtype)
mask[0] = 1.0
eps_z = torch.randn(T, B, n).to(dtype) if sample else None
eps_a = torch.randn(T, B, p).to(dtype) if sample else None
cots = None
grads = {}
for dt in (torch.float64, torch.float32):
    c = lambda t: None if t is None else t.to(dt)
    f1, l1 = torch.nn.Linear(p, H).to(dt), torch.nn.Linear(H, K).to(dt)
    f1.load_state_dict({k: v.to(dt) for k, v in first.state_dict().items()})
    l1.load_state_dict({k: v.to(dt) for k, v in last.state_dict().items()})
    leaves = [c(t).requires_grad_(True) for t in (a, A_K, C_K, Q, R, m0, P0)]
    net = lambda x: torch.softmax(l1(torch.relu(f1(x))), dim=-1)
    out = O.kalman_filter(*leaves, net, mask=c(mask), eps_state=c(eps_z), eps_obs=c(eps_a),
                          learn_weight_model=True, burn_in=burn_in)
    if cots is None:
        cots = {k: torch.randn_like(out[k]) for k in NAMES}
    loss = sum((cots[k].to(dt) * out[k]).sum() for k in NAMES)
    grads[dt] = torch.autograd.grad(loss, leaves + [f1.weight, f1.bias, l1.weight, l1.bias])
names = ["a", "A_K", "C_K", "Q", "R", "m0", "P0", "W1", "b1", "W2", "b2"]
print("float32 oracle against float64 oracle:",
      ", ".join(f"{k} {rel_err(g32, g64):.2e}" for k, g32, g64 in zip(names, grads[torch.float32], grads[torch.float64])))
