#!/usr/bin/env python3
"""Randomised parity sweep of the HIP library against the float64 oracle: random T, B,
dimensions, masks, options, both small-state families and both large-state families, the
one-launch masked filter with the perceptron network inside (forward, and under autograd),
the fused objective and the one-launch filter + smoother.
Uses the same checks as the GPU tests.  usage: stress_parity.py [cases] [seed]"""
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kernel_checks as KC  # noqa: E402
import test_gpu_fused_inference as TFI  # noqa: E402
import test_gpu_masked_grad as TMG  # noqa: E402
from kernel_harness import Backend  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SMALL = [(4, 2, 3), (2, 1, 2), (3, 2, 2), (8, 4, 4), (4, 2, 8)]
ODD = [(5, 3, 4), (4, 2, 5), (3, 1, 1), (7, 4, 2), (2, 2, 3)]     # run as a block of a larger problem
LARGE = [(32, 16, 8), (16, 8, 4)]
self_check = bool(os.environ.get("STRESS_SELF_CHECK"))
if self_check:                               # the sweep must notice errors: impossible tolerance
    KC.TOL[torch.float32] = 1e-9
be, host = Backend("hip"), Backend("hostsim")
failures, conditioning, t0 = [], [], time.time()
for i in range(cases):
    large = rng.random() < 0.3
    n, p, K = rng.choice(LARGE if large else SMALL)
    dtype = torch.float32 if (large or rng.random() < 0.6) else torch.float64
    T = rng.choice([1, 2, 3, 5, 8, 13, 20, 31])
    B = rng.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 130, 257]) if not large else rng.choice([1, 2, 5, 9])
    # 16: the fused objective with one sequence over 16 lanes (where it exists: float, n and p up
    # to 4, T <= 35); the library takes the quad form for the other calls and shapes
    os.environ["KVAE_SEQ_LANES"] = rng.choice(["1", "4", "16"])
    os.environ["KVAE_LARGE_STATE"] = rng.choice(["tile", "lds"])
    sym, burn = rng.random() < 0.7, rng.choice([0, 0, 1, 3])
    burn = min(burn, max(T - 1, 0))
    kind = rng.choice(["filter", "smooth", "elbo"]
                      + ([] if large else ["masked_mlp", "loss", "masked_grad", "fused_inference"]))
    desc = (f"{kind} T={T} B={B} n={n} p={p} K={K} {str(dtype)[6:]} lanes={os.environ['KVAE_SEQ_LANES']} "
            f"large={os.environ['KVAE_LARGE_STATE']} sym={sym} burn_in={burn}")
    try:
        if kind == "filter":
            filter_args = dict(T=T, B=B, n=n, p=p, K=K, dtype=dtype, seed=1000 + i,
                               mask_kind=rng.choice(["none", "binary", "fractional"]),
                               symmetrize=sym, burn_in=burn,
                               per_sequence_carry=rng.random() < 0.3)
            KC.check_filter(be, **filter_args)
        elif kind == "masked_mlp":
            KC.check_masked_mlp("cuda:0", T=T, B=B, n=n, p=p, K=K,
                                hidden=rng.choice([1, 7, 32, 128, 200]), dtype=dtype, seed=4000 + i,
                                mask_kind=rng.choice(["none", "binary", "fractional"]),
                                sample=rng.random() < 0.5, symmetrize=sym)
        elif kind == "loss":
            loss_args = dict(T=max(T, 2), B=B, n=n, p=p, K=K, dtype=dtype, seed=5000 + i,
                             mask_kind=rng.choice(["none", "binary"]), symmetrize=sym, burn_in=burn)
            KC.check_loss(be, **loss_args)
        elif kind == "masked_grad":
            no, po, Ko = rng.choice(SMALL + ODD)
            mg_args = dict(sample=rng.random() < 0.5, burn_in=burn, learn=rng.random() < 0.8,
                           T=max(T, 2), B=B, H=rng.choice([1, 7, 32, 128]), seed=6000 + i)
            desc += f" (dims {no},{po},{Ko}; {mg_args})"
            TMG.run_case(no, po, Ko, dtype, **mg_args)
        elif kind == "fused_inference":
            no, po, Ko = rng.choice(SMALL + ODD)
            from kalman_vae_amd import functions as F
            d = TFI.problem(max(T, 2), B, no, po, Ko, dtype, 7000 + i, rng.random() < 0.5)
            sample = d["eps_state"] is not None
            alpha = torch.softmax(torch.randn(max(T, 2) + 1, B, Ko, dtype=dtype, device="cuda:0"), -1)
            with torch.no_grad():
                filt = F.kalman_filter_steps(d["a"], alpha, d["mask"], d["A_K"], d["C_K"], d["Q"],
                                             d["R"], d["m0"], d["P0"], symmetrize=sym)
                smo = F.kalman_smooth_steps(*filt, LR=d["L_R"] if sample else None,
                                            eps_state=d["eps_state"], eps_obs=d["eps_obs"],
                                            symmetrize=sym)
            out = F.filter_smooth_fused(d["a"], d["mask"], d["A_K"], d["C_K"], d["Q"], d["R"],
                                        d["m0"], d["P0"], alpha=alpha, L_R=d["L_R"],
                                        eps_state=d["eps_state"], eps_obs=d["eps_obs"],
                                        symmetrize=sym, want=TFI.NAMES)
            tol = 5e-6 if dtype == torch.float32 else 1e-12
            for k, want in zip(TFI.NAMES, tuple(filt) + tuple(smo)):
                e = KC.rel_err(out[k], want)
                assert e < tol, f"{k}: {e:.2e}"
            desc += f" (dims {no},{po},{Ko})"
        elif kind == "smooth":
            smooth_args = dict(T=T, B=B, n=n, p=p, K=K, dtype=dtype, seed=2000 + i,
                               sample=rng.random() < 0.7, symmetrize=sym, burn_in=burn)
            KC.check_smooth(be, **smooth_args)
        else:
            KC.check_elbo(be, T=T, B=B, n=n, p=p, K=K, dtype=dtype, seed=3000 + i)
    except Exception as err:  # noqa: BLE001
        msg = desc + " :: " + str(err).replace("\n", " | ")[:300]
        # A float32 case whose terms cancel (e.g. the 1x1 d_R of p = 1) can miss 1e-4 in any
        # summation order: the same source compiled for the host, summing sequentially,
        # then misses it too.  Such cases are listed apart; a kernel defect would not repeat
        # on the host build.
        alike = False
        if dtype == torch.float32 and kind != "masked_mlp" and not large and not self_check:
            try:
                args = dict(T=T, B=B, n=n, p=p, K=K, dtype=dtype)
                if kind == "elbo":
                    KC.check_elbo(host, seed=3000 + i, **args)
                elif kind == "loss":
                    KC.check_loss(host, **loss_args)
                elif kind == "filter":
                    KC.check_filter(host, **filter_args)
                elif kind == "smooth":
                    KC.check_smooth(host, **smooth_args)
            except Exception:  # noqa: BLE001
                alike = True
        if dtype == torch.float32 and kind == "masked_grad" and not self_check:
            # no host build of this path takes float32 operands: the same case in float64
            # (1e-8) tells a float32 rounding miss from a defect of the kernel
            try:
                TMG.run_case(no, po, Ko, torch.float64, **mg_args)
                alike = True
                msg += "  [the same case in float64 meets 1e-8]"
            except Exception:  # noqa: BLE001
                pass
        (conditioning if alike else failures).append(msg)
    if (i + 1) % 25 == 0:
        print(f"{i + 1} cases, {len(failures)} failures, {time.time() - t0:.0f} s", flush=True)
for msg in conditioning:
    print("float32 conditioning (the host build of the same source misses the tolerance alike, or the "
          "float64 kernels meet theirs): " + msg)
print("\n".join(failures) if failures else "all cases within tolerance"
      + (f" except {len(conditioning)} ill-conditioned float32 case(s) listed above" if conditioning else ""))
sys.exit(1 if failures else 0)
