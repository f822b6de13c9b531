#!/usr/bin/env python3
"""Gate for the 4-row tail packing of k_sparse_conv: the histogram of (block, offset) list lengths
mod 16 per octree level of the bench cloud, and the tile fill it implies for

  * 16-row tiles (what k_sparse_conv issues today:  fill = pairs / (16 * sum ceil(len / 16))), and
  * full 16-row tiles + tails packed 4 rows at a time on v_mfma_f32_4x4x1_16B_f32
    (fill4 = pairs / sum(16 * floor(len / 16) + 4 * ceil((len mod 16) / 4))),

computed on the CPU from the cloud alone (no GPU, no library): level coordinates = xyz >> d, rows in Morton order,
blocks of H consecutive rows (H = 255: the encoder's class; the decoder's balanced heights are passed with --heights).
Also prints what share of the pairs would go through the tail path -- that share pays 4 KiB of weights per 4 rows.

Usage: tools/tile_fill_model.py [points] [--k 5] [--H 255]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gauspcc_amd.synth import synthetic_cloud  # noqa: E402


def part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def level_stats(c, k, H):
    """c: (n, 3) unique int64 coords of one level.  Returns (n, pairs, lens) with lens = the (block, offset) list lengths > 0."""
    n = c.shape[0]
    mk = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << np.uint64(1)) | (part1by2(c[:, 2]) << np.uint64(2))
    order = np.argsort(mk, kind="stable")
    c = c[order]
    key = (c[:, 2] << 42) | (c[:, 1] << 21) | c[:, 0]
    ks = np.sort(key)
    blk = np.arange(n, dtype=np.int64) // H
    nblk = int(blk[-1]) + 1
    r = k // 2
    lens = []
    pairs = 0
    for dz in range(-r, r + 1):
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                q = key + ((dz << 42) + (dy << 21) + dx)
                ok = (c[:, 0] + dx >= 0) & (c[:, 1] + dy >= 0) & (c[:, 2] + dz >= 0)
                pos = np.searchsorted(ks, q)
                pos[pos >= n] = n - 1
                hit = ok & (ks[pos] == q)
                cnt = np.bincount(blk[hit], minlength=nblk)
                pairs += int(hit.sum())
                lens.append(cnt[cnt > 0])
    return n, pairs, np.concatenate(lens)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("points", nargs="?", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--H", type=int, default=255)
    ap.add_argument("--min-nodes", type=int, default=16385, help="levels below this run the cooperative / fused kernels, not k_sparse_conv")
    a = ap.parse_args()
    xyz = synthetic_cloud(a.points, seed=1234).astype(np.int64)
    xyz -= xyz.min(axis=0)
    tot = dict(pairs=0, t16=0, t4=0, tail_pairs=0, runs=0)
    print(f"# tools/tile_fill_model.py {a.points} --k {a.k} --H {a.H}: (block, offset) list lengths of the bench cloud, per level (d = shift from the leaves)")
    print("# level      nodes  pairs/row  runs  len/run | len mod 16 histogram (share of runs: 0, 1-4, 5-8, 9-12, 13-15) | fill16  fill16+4  tail share of pairs | weight KiB per pair now -> with tails")
    d = 0
    while True:
        c = np.unique(xyz >> d, axis=0)
        if c.shape[0] < a.min_nodes:
            break
        # the coded levels are d >= 1 (the leaves themselves carry no features); d = 0 is listed for the histogram only
        n, pairs, lens = level_stats(c, a.k, a.H)
        m = lens % 16
        full = lens // 16
        t16 = int(np.ceil(lens / 16).sum())
        t4_rows = int((16 * full + 4 * np.ceil(m / 4)).sum())
        tail_pairs = int(m.sum())
        h = [np.mean(m == 0), np.mean((m >= 1) & (m <= 4)), np.mean((m >= 5) & (m <= 8)), np.mean((m >= 9) & (m <= 12)), np.mean(m >= 13)]
        w_now = 4.0 * t16 / pairs                                     # one 4 KiB fragment per tile (per pair of tiles where paired: halve it)
        w_new = (4.0 * full.sum() + 4.0 * np.ceil(m / 4).sum()) / pairs  # 4 KiB per full tile + 4 KiB per 4-row group
        print(f"  d={d:2d} {n:10d} {pairs / n:9.1f} {len(lens):8d} {lens.mean():7.1f} | " + " ".join(f"{v:5.2f}" for v in h)
              + f" | {pairs / (16 * t16):6.3f} {pairs / t4_rows:8.3f} {tail_pairs / pairs:10.3f} | {w_now:6.2f} -> {w_new:6.2f}")
        if d >= 1:
            tot["pairs"] += pairs; tot["t16"] += t16; tot["t4"] += t4_rows; tot["tail_pairs"] += tail_pairs; tot["runs"] += len(lens)
        d += 1
    print(f"# coded levels (d >= 1), weighted by pairs: fill16 {tot['pairs'] / (16 * tot['t16']):.3f}  fill16+4 {tot['pairs'] / tot['t4']:.3f}  "
          f"tail share of pairs {tot['tail_pairs'] / tot['pairs']:.3f}  MFMA rows issued {16 * tot['t16']} -> {tot['t4']} ({tot['t4'] / (16 * tot['t16']) - 1:+.1%})")


if __name__ == "__main__":
    main()
