#!/usr/bin/env python3
"""Per decode level (between k_expand launches of the last decode in a rocprofv3 kernel trace): span, kernels, busy time by kernel class."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gpcc::", "")))
rows.sort()
ex = [i for i, r in enumerate(rows) if r[2].startswith("k_expand")]
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 14
ex = ex[-nl:] + [len(rows)]
for li in range(nl):
    a, b = ex[li], ex[li + 1]
    seg = rows[a:b]
    span = (seg[-1][1] - seg[0][0]) / 1e3
    cls = collections.Counter()
    for s, e, n in seg:
        key = "conv" if "conv" in n else "rc" if "k_rc_" in n else "head" if "head" in n else "tiles" if ("tile" in n or "chunk_tiles" in n or "block_sum" in n) else "elem" if ("stage_input" in n or "embed" in n or "child_feat" in n or "assemble" in n) else "other"
        cls[key] += (e - s) / 1e3
    print(f"level {li:2d}: kernels {b - a:3d} span {span:8.1f} us  " + "  ".join(f"{k} {v:7.1f}" for k, v in sorted(cls.items())))
