#!/usr/bin/env python3
"""Median per-launch value of each SQ counter per kvae kernel from rocprofv3 --pmc csv files.
usage: pmc_sq.py counters.csv [...]"""
import collections
import csv
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "kvae::" not in name:
            continue
        import re
        short = re.sub(r"\(anonymous namespace\)::", "", name.split("kvae::", 1)[1]).split("(")[0]
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(vals.items()):
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    waves = med.get("SQ_WAVES", 0.0)
    print(f"{k}: {max(len(v) for v in c.values())} launches, {int(waves)} wavefronts")
    for n, v in sorted(med.items()):
        per = f"  per wavefront {v / waves:12.0f}" if waves else ""
        print(f"   {n:34s} {v:16.0f}{per}")
