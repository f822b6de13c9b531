#!/usr/bin/env python3
"""Per-grid-size summary of k_sparse_conv dispatches from a rocprofv3 kernel trace CSV."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "k_sparse_conv" not in r["Kernel_Name"]:
            continue
        g = (int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]))
        rows[g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f"total conv us {tot:.0f}")
print("blocks jobs  calls   avg_us   sum_us  frac")
for g in sorted(rows):
    v = rows[g]
    print(f"{g[0]:6d} {g[1]:4d} {len(v):6d} {sum(v)/len(v):8.1f} {sum(v):8.0f} {sum(v)/tot:5.3f}")
