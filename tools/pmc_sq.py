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
        short = name.split("kvae::")[1].split("(")[0]
        if "loss_row_kernel" not in short:      # the row kernels differ by template arguments
            short = short.split("<")[0]
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in vals.items():
    print(k, {n: sorted(v)[len(v) // 2] for n, v in c.items()})
