#!/usr/bin/env python3
"""Per-level breakdown of the LAST decode in a rocprofv3 kernel trace: a level starts at the k_embed_occ of its parent
trunk (or at the fused level kernel); per level the span, the time no kernel runs, and kernel time by category.
Usage: tools/level_breakdown.py trace.csv [queue-aware]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gpcc::", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
rows.sort()
# the last decode: what follows an encode's k_rc_compact up to the next encode's first kernel (k_bbox), with at least 3 trunks
cuts = [i for i, r in enumerate(rows) if "k_rc_compact" in r[2]]
starts = [i for i, r in enumerate(rows) if "k_bbox" in r[2]] + [len(rows)]
dec = []
for c in reversed(cuts):
    nxt = min(i for i in starts if i > c)
    cand = rows[c + 1: nxt]
    if sum(1 for r in cand if r[2] == "k_embed_occ" or "k_level_fused" in r[2]) >= 3:
        dec = cand
        break
for i in range(len(dec) - 1):
    if dec[i + 1][0] - dec[i][1] > 20e6:
        dec = dec[: i + 1]
        break


def cat(n):
    if "sparse_conv" in n or "conv_products" in n or "conv_sum" in n or "k_level_fused" in n:
        return "conv"
    if "rc_decode" in n or "k_rc_" in n or "cp_decode" in n:
        return "coder"
    if "head" in n:
        return "head"
    if n in ("k_embed_occ", "k_child_features", "k_stage_input_dec", "k_assemble_occ"):
        return "elem"
    if "tiles" in n or "k_block_sum" in n or "k_tile_words" in n or "k_fold_pairs" in n or "k_pad_tiles" in n:
        return "tiles"
    if "copyBuffer" in n or "fillBuffer" in n:
        return "copy"
    return "octree"


marks = [i for i, r in enumerate(dec) if r[2] == "k_embed_occ"]
marks.append(len(dec))
print(f"decode: {len(dec)} kernels, span {(dec[-1][1] - dec[0][0]) / 1e3:.1f} us; head (before the first trunk) {(dec[marks[0]][0] - dec[0][0]) / 1e3:.1f} us")
print(f"{'lvl':>3} {'kern':>4} {'span':>8} {'idle':>7} | " + " ".join(f"{c:>8}" for c in ("conv", "coder", "head", "elem", "tiles", "octree", "copy")) + " | launches by category")
tot = collections.Counter()
for li in range(len(marks) - 1):
    seg = dec[marks[li]: marks[li + 1]]
    t0, t1 = seg[0][0], (dec[marks[li + 1]][0] if marks[li + 1] < len(dec) else seg[-1][1])
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    by = collections.Counter()
    cnt = collections.Counter()
    for s, e, n, _ in seg:
        by[cat(n)] += e - s
        cnt[cat(n)] += 1
    tot.update(by)
    tot["span"] += t1 - t0
    tot["idle"] += (t1 - t0) - busy
    tot["kern"] += len(seg)
    print(f"{li:3d} {len(seg):4d} {(t1 - t0) / 1e3:8.1f} {((t1 - t0) - busy) / 1e3:7.1f} | " + " ".join(f"{by[c] / 1e3:8.1f}" for c in ("conv", "coder", "head", "elem", "tiles", "octree", "copy"))
          + " | " + " ".join(f"{c}:{cnt[c]}" for c in ("conv", "coder", "head", "elem", "tiles", "octree", "copy")))
print(f"sum {tot['kern']:4d} {tot['span'] / 1e3:8.1f} {tot['idle'] / 1e3:7.1f} | " + " ".join(f"{tot[c] / 1e3:8.1f}" for c in ("conv", "coder", "head", "elem", "tiles", "octree", "copy")))
