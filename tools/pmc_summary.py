#!/usr/bin/env python3
"""Combine rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kB units) into HBM
bytes per launch per kernel and compare with the kernels' algorithmic bytes.
usage: pmc_summary.py T B n p K fetch.csv write.csv"""
import collections
import csv
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

T, B, n, p, K = map(int, sys.argv[1:6])
alg = bench.kernel_bytes(n, p, K, 4)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[6:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "kvae::" not in name:
            continue
        short = name.split("kvae::")[1].split("<")[0].replace("_kernel", "").replace("wave_", "").replace("_quad", "")
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | FETCH_SIZE kB (raw) | WRITE_SIZE kB | algorithmic read+write MB | "
      "counted MB (fetch x2 + write) | counted / algorithmic |")
print("|---|---:|---:|---:|---:|---:|")
for k, c in vals.items():
    if k not in alg:
        continue
    f = sorted(c.get("FETCH_SIZE", [0]))[len(c.get("FETCH_SIZE", [0])) // 2]
    w = sorted(c.get("WRITE_SIZE", [0]))[len(c.get("WRITE_SIZE", [0])) // 2]
    a_mb = alg[k] * T * B / 1e6
    counted = (2 * f + w) * 1024 / 1e6
    print(f"| {k} | {f:.0f} | {w:.0f} | {a_mb:.1f} | {counted:.1f} | {counted / a_mb:.2f} |")
