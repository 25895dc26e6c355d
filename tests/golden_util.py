"""Helpers shared by the tests: load golden fixtures (tests/golden/*.npz, made
by tests/golden/make_golden.py from the reference) and rebuild their inputs."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(pattern="*"):
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, pattern + ".npz")))


class Golden:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.cfg = json.loads(str(z["config"]))
        self.dtype = torch.float64 if self.cfg["dtype"] == "float64" else torch.float32
        self.t = {k: torch.from_numpy(np.array(z[k])) for k in z.files if k != "config"}
        # "store_f32" fixtures were computed in float64 and stored in float32
        self.stored_f32 = bool(self.cfg.get("store_f32", False))
        self.t = {k: (v.to(self.dtype) if v.is_floating_point() else v)
                  for k, v in self.t.items()}
        # large parameter tensors that the fixture regenerates from a seed instead of storing
        for k, (shape, bound) in self.cfg.get("seeded", {}).items():
            self.t["sd." + k] = seeded_tensor(k, shape, self.cfg["seed"], bound).to(self.dtype)

    def __getitem__(self, key):
        return self.t[key]

    def has(self, key):
        return key in self.t

    def weight_model(self):
        """The package's weight network with the fixture's parameters."""
        from kalman_vae_amd.dynamics_parameter_network import make_weight_model
        c = self.cfg
        net = make_weight_model(c["net"], c["p"], c["K"], c.get("hidden_dim", 8),
                                c.get("num_layers", 2)).to(self.dtype)
        sd = {k[len("sd.weight_model."):]: v for k, v in self.t.items()
              if k.startswith("sd.weight_model.")}
        net.load_state_dict(sd)
        return net

    def noise(self, which):
        """(eps_state, eps_obs) for 'filter' / 'smooth' or None where the
        fixture ran in "mean" mode."""
        st = self.cfg["state_transition"] == "sample"
        ob = self.cfg["observation"] == "sample"
        return (self.t[f"eps.{which}_state"] if st else None,
                self.t[f"eps.{which}_obs"] if ob else None)


def grad_mismatch(g, key, got, tol):
    """Relative error of gradient ``got`` against fixture ``g``'s record of parameter ``key``:
    the whole tensor, or (seeded parameters) every 16th element and the sum.  None when the
    fixture holds no gradient for it."""
    if g.has("grad." + key):
        return rel_err(got, g["grad." + key])
    if g.has("gradsub." + key):
        flat = got.detach().reshape(-1)
        e_sub = rel_err(flat[::16], g["gradsub." + key])
        # the sum against the size of its terms (it cancels)
        e_sum = float(abs(flat.double().sum().cpu() - g["gradsum." + key].double())
                      / flat.double().abs().sum().cpu().clamp_min(1e-30))
        return max(e_sub, e_sum)
    return None


def rel_err(x, ref):
    x = x.detach().double().cpu()
    ref = ref.detach().double().cpu()
    scale = ref.abs().max().clamp_min(1e-30)
    return float((x - ref).abs().max() / scale)


def seeded_tensor(name, shape, seed, bound):
    """A float32-representable tensor that depends only on (name, shape, seed): uniform in
    (-bound, bound) from numpy's frozen legacy generator.  Lets a fixture leave out large
    parameter tensors: the script that made it and the test that replays it both call
    this."""
    import zlib
    rs = np.random.RandomState((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
    v = rs.uniform(-bound, bound, size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(v)
