#!/usr/bin/env python3
"""Gap analysis of a rocprofv3 kernel trace: for the last encode+decode step, total kernel-busy time vs. idle
gaps between consecutive kernels (launch-bound stretches)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# steps are separated by k_bbox (first kernel of an encode)
starts = [i for i, r in enumerate(rows) if "k_bbox" in r[2]]
lo = starts[-1]
seg = rows[lo:]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps = [(seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
print(f"last step: {len(seg)} kernels, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {sum(g for g in gaps if g>0)/1e6:.2f} ms")
big = sorted(((g, seg[i][2], seg[i + 1][2]) for i, g in enumerate(gaps)), reverse=True)[:25]
for g, a, b in big:
    print(f"  gap {g/1e3:8.1f} us  after {a[:60]}  before {b[:60]}")
import collections
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    agg[n][0] += 1; agg[n][1] += e - s
print("per kernel (last step):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {t/1e3:9.1f} us {c:5d}  {n[:100]}")
