"""One case of the randomised parity sweep (fused objective, z_dim 2, a_dim 1, K 2, T 2, B 130):
d_R of the HIP library in its lane forms (16 lanes per sequence: float only) and both precisions
against the float64 oracle, with the relative error of d_R itself."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_checks as KC  # noqa: E402
from kernel_harness import Backend  # noqa: E402

be, host = Backend("hip"), Backend("hostsim")
args = dict(T=2, B=130, n=2, p=1, K=2, seed=5000 + 1245, mask_kind="none", symmetrize=True, burn_in=0)
for dtype in (torch.float32, torch.float64):
    for lanes in ("1", "4", "16"):
        if lanes == "16" and dtype != torch.float32:
            continue
        os.environ["KVAE_SEQ_LANES"] = lanes
        for name, b in (("hip", be), ("host", Backend("hostsim_row") if lanes == "16" else host)):
            try:
                KC.check_loss(b, dtype=dtype, **args)
                print(dtype, "lanes", lanes, name, "ok")
            except AssertionError as e:
                print(dtype, "lanes", lanes, name, "FAIL", str(e).replace("\n", " | ")[:200])
# the size of d_R against the sum of the magnitudes of its per-sequence parts
prob = KC.random_problem(2, 130, 2, 1, 2, torch.float32, 5000 + 1245, "none")
print("R", prob["R"].flatten().tolist(), "Q", prob["Q"].flatten().tolist())
