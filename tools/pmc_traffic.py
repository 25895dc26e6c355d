#!/usr/bin/env python3
"""Median HBM bytes per launch per kernel from rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected in separate runs; both count kB; FETCH_SIZE is doubled: gfx950 reports
half the bytes of wide coalesced reads).  usage: pmc_traffic.py counter_collection.csv ..."""
import collections
import csv
import json
import re
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "kvae::" not in name:
            continue
        # "void kvae::(anonymous namespace)::kernel<1, 2>(args...)" -> "kernel<1, 2>"
        short = re.sub(r"\(anonymous namespace\)::", "", name.split("kvae::", 1)[1]).split("(")[0]
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
out = {}
for k, c in sorted(vals.items()):
    f, w = med(c.get("FETCH_SIZE", [])), med(c.get("WRITE_SIZE", []))
    out[k] = {"launches": max(len(v) for v in c.values()), "fetch_kB_raw": f, "write_kB": w,
              "hbm_bytes": (2 * f + w) * 1024}
print(json.dumps(out, indent=1))
