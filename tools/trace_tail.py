"""Print the kernel sequence (name, queue, start) of a rocprofv3 kernel trace between two markers."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# find the last two occurrences of filter_forward (start of a step)
idx = [i for i, n in enumerate(names) if "filter_forward_kernel" in n]
print("steps at", idx)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
lo = idx[which]
hi = idx[which + 1] if which + 1 < 0 and which + 1 < len(idx) else len(rows)
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} q{r["Queue_Id"]} s{r.get("Stream_Id", "?")} {r["Kernel_Name"][:110]}')
