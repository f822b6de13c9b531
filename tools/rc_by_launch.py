#!/usr/bin/env python3
"""Per-launch durations of the range-coder kernels in a rocprofv3 kernel trace (last encode + decode step).
Usage: tools/rc_by_launch.py kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("LDS_Block_Size", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_bbox" in r[2]]
seg = rows[starts[-1]:]
tot = {}
for s, e, n, g, l in seg:
    if "k_rc_" in n:
        short = n.split("(")[0].replace("gpcc::", "").replace("void ", "")
        print(f"{short:34s} grid {g:>8s} lds {l:>6s}  {(e - s) / 1e3:8.1f} us")
        tot[short] = tot.get(short, 0) + (e - s) / 1e3
print({k: round(v, 1) for k, v in tot.items()})
