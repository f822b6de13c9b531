#!/usr/bin/env python3
"""Kernel sequence of one decode level (between two k_expand launches) from a rocprofv3 kernel trace."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpcc::", "")))
rows.sort()
ex = [i for i, r in enumerate(rows) if r[2].startswith("k_expand")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = ex[-15 + which], ex[-15 + which + 1]
t0 = rows[a][0]
for s, e, n in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n[:70]}")
print("kernels", b - a, "span us", (rows[b][0] - t0) / 1e3)
