"""Random sweep of the masked filter with the LSTM network as one autograd node against autograd
through the float64 oracle (tests/test_gpu_masked_lstm.py::run_case on random shapes and
options).  python tools/exp/sweep_masked_lstm.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_masked_lstm as M  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for i in range(cases):
    n, p, K = rng.choice([(4, 2, 3), (2, 1, 2), (3, 2, 2), (4, 2, 3)])
    kw = dict(sample=rng.random() < 0.6, burn_in=rng.choice([0, 0, 1, 3]), learn=rng.random() < 0.8,
              T=rng.choice([1, 2, 3, 5, 8, 13]), B=rng.choice([1, 2, 15, 16, 17, 33, 64, 130]),
              seed=rng.randrange(10 ** 6), exact=rng.random() < 0.3)
    try:
        M.run_case(n, p, K, **kw)
    except AssertionError as e:
        bad += 1
        print("FAIL", (n, p, K), kw, str(e).replace("\n", " | ")[:300], flush=True)
print(f"{cases} cases, {bad} failures")
