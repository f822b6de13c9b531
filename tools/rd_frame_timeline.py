#!/usr/bin/env python3
"""Kernel sequence of the LAST RD frame in a rocprofv3 kernel trace of tools/rd_frame_probe.py: start, duration, gap to the previous kernel's end."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
ren = [i for i, r in enumerate(rows) if r[2].startswith("k_render")]
a = ren[-2] + 1 if len(ren) > 1 else 0
seg = rows[a:ren[-1] + 1]
t0 = seg[0][0]
end = seg[0][0]
busy = 0
for s, e, n in seg:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {max(0, s - end) / 1e3:7.1f}  {n[:80]}")
    busy += e - s
    end = max(end, e)
print(f"kernels {len(seg)}  span {(end - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us")
