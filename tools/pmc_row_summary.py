#!/usr/bin/env python3
"""Per-wavefront medians of the SQ counters of the loss_row_kernel instantiations from the csv
files tools/pmc_row.sh leaves under gpurun_out/pmc_row/.  usage: pmc_row_summary.py [dir]"""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_row"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/p*/p*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "loss_row_kernel<" not in n:
            continue
        key = n.split("loss_row_kernel")[1].split(">")[0] + ">"
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in vals.items():
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    W = med["SQ_WAVES"]
    print(k, "launches", len(next(iter(c.values()))), "waves", int(W))
    for n, v in sorted(med.items()):
        print(f"   {n:24s} {v:14.0f}  per wave {v / W:10.0f}")
