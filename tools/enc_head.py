#!/usr/bin/env python3
"""Developer probe: the kernels of the last encode in a rocprofv3 kernel trace up to its first convolution."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("gpcc::", "").replace("(anonymous namespace)::", "").split("(")[0]))
rows.sort()
lk = [i for i, r in enumerate(rows) if r[2].startswith("k_leaf_keys")]
a = lk[-1]
while a > 0 and rows[a][0] - rows[a - 1][1] < 30000: a -= 1
t0 = rows[a][0]
for s, e, n in rows[a:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n[:90]}")
    if n.startswith("k_sparse_conv"): break
