#!/usr/bin/env python3
"""Gap analysis of a rocprofv3 kernel trace: for the last encode+decode step, kernel-busy time vs idle gaps,
separately for the encode (k_bbox .. k_rc_compact) and the decode (the rest)."""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_bbox" in r[2] or "k_fbbox" in r[2]]   # (k_fbbox: the first kernel of a batched encode)
seg = rows[starts[-1]:]
cut = max(i for i, r in enumerate(seg) if "k_rc_compact" in r[2]) + 1


def report(name, seg):
    busy = sum(e - s for s, e, _ in seg)
    span = max(e for _, e, _ in seg) - seg[0][0]
    # Kernels of several streams overlap: a gap is a stretch that NO kernel covers (round 5: until then a gap was the distance between
    # consecutive kernels in start order, which reported a kernel of the second stream that starts and ends under a long convolution as
    # a hole in front of the next convolution -- the "308 us hole before the first trunk" of round 4's encode was one of those).
    gaps, cover_end, cover_name = [], seg[0][1], seg[0][2]
    for s_, e_, n_ in seg[1:]:
        gaps.append((s_ - cover_end, cover_name, n_))
        if e_ > cover_end:
            cover_end, cover_name = e_, n_
    idle = sum(g for g, _, _ in gaps if g > 0)
    print(f"== {name}: {len(seg)} kernels, span {span/1e6:.2f} ms, busy (summed over streams) {busy/1e6:.2f} ms, idle (no kernel running) {idle/1e6:.2f} ms")
    hist = collections.Counter()
    for g, _, _ in gaps:
        b = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"
        hist[b] += max(g, 0)
    print("   idle by gap size:", {k: f"{v/1e6:.2f}ms" for k, v in hist.items()})
    for g, a, b in sorted(gaps, reverse=True)[:8]:
        print(f"   gap {g/1e3:8.1f} us  after {a[:50]}  before {b[:50]}")
    big = sorted(range(len(gaps)), key=lambda i: -gaps[i][0])[:3]
    for i in sorted(big):
        print(f"   around the {gaps[i][0]/1e3:.0f} us gap (t = {(seg[i][1] - seg[0][0])/1e3:.0f} us after the first kernel):")
        for s_, e_, n_ in seg[max(0, i - 3): i + 5]:
            print(f"      {(s_ - seg[0][0])/1e3:9.1f} us  +{(e_ - s_)/1e3:7.1f}  {n_[:60]}")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        agg[n][0] += 1
        agg[n][1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"   {t/1e3:9.1f} us {c:5d}  {n[:90]}")


report("encode", seg[:cut])
import os
if os.environ.get("TL_WINDOW"):   # every kernel of the encode inside [a, b] us after its first kernel
    a, b = (float(v) * 1e3 for v in os.environ["TL_WINDOW"].split(","))
    for s_, e_, n_ in seg[:cut]:
        if a <= s_ - seg[0][0] <= b:
            print(f"      {(s_ - seg[0][0])/1e3:9.1f} us  +{(e_ - s_)/1e3:7.1f}  {n_[:70]}")
dec = seg[cut:]
# drop anything after the decode (teardown): cut at the first gap longer than 20 ms
last = len(dec) - 1
for i in range(len(dec) - 1):
    if dec[i + 1][0] - dec[i][1] > 20e6:
        last = i
        break
report("decode", dec[: last + 1])
print(f"== gap between the encode's last kernel and the decode's first: {(dec[0][0] - seg[cut - 1][1]) / 1e6:.2f} ms")
