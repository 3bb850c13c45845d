"""Per-launch durations of a kernel from a rocprofv3 --kernel-trace csv (in dispatch order)."""
import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "k_sort_level"
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
for s, e, n, g in rows:
    print(f"{(e - s) / 1e6:8.3f} ms  grid {g:>10}  {n}")
print("total %.3f ms over %d launches" % (sum(e - s for s, e, _, _ in rows) / 1e6, len(rows)))
