#!/usr/bin/env python3
"""Developer probe: the first / last N kernels of the last decode in a rocprofv3 kernel trace (start offsets in us)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpcc::", "").replace("(anonymous namespace)::", "")))
rows.sort()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
enc = [i for i, r in enumerate(rows) if r[2].startswith("k_rc_encode")]
a = enc[-1]
t0 = rows[a][1]
print("== after the last k_rc_encode")
for s, e, n in rows[a:a + N]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n[:70]}")
print("== end of the trace")
for s, e, n in rows[-25:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  {n[:70]}")
