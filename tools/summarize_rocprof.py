#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel names are
cut to 90 characters) so that profile summaries stay readable in profiles/."""
import csv
import sys


def main(path, steps=None):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {path}")
    print(f"total kernel time {total/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches"
          + (f"; {total/1e6/steps:.3f} ms per bench step ({steps} steps incl. warm-up)" if steps else ""))
    print()
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows:
        name = r["Name"].replace("|", "/")
        if len(name) > 90:
            name = name[:87] + "..."
        print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
