#!/usr/bin/env python3
"""Soak of the batched codec (gpcc_encode_batch / gpcc_decode_batch) on random batch shapes: K scenes of random sizes (1 .. 300 k points,
log-uniform), random extents (2^6 .. 2^17: depths differ inside a batch, some scenes are a base level only), random offsets in space (negative
coordinates included), random kernel size (3 / 5) and lane length.  Every iteration checks
    * every scene's batched bytes == its solo bytes (gpcc_encode through _encode_to_bytes),
    * the batched decode of those bytes == the solo decode (same points, same order),
    * the decoded set == the input set.
Prints one line per iteration and a summary; exit code 1 on the first mismatch.
    python tools/batch_soak.py SECONDS [SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gauspcc_amd import runtime  # noqa: E402
from gauspcc_amd.pcc_utils import _decode_batch, _decode_bytes, _encode_batch, _encode_to_bytes  # noqa: E402
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
dev = torch.device("cuda", 0)
models = {k: runtime.Model(synthetic_state_dict(32, k), 32, k, 0) for k in (3, 5)}


def lex(a):
    return a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))]


def scene(i):
    n = int(np.exp(rng.uniform(0, np.log(300_000))))
    ext = int(rng.randint(6, 18))
    while (1 << (3 * ext)) < 4 * n:      # room for n distinct voxels
        ext += 1
    if rng.rand() < 0.15:                # a dense little block: a tree of two or three levels
        side = max(2, int(round(n ** (1 / 3))) + 1)
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3)
        pts = g[rng.permutation(len(g))[: max(1, min(n, len(g)))]].astype(np.int32)
    else:
        pts = synthetic_cloud(max(n, 1), seed=int(rng.randint(1 << 30)), extent_log2=ext)
    shift = rng.randint(-(1 << 18), 1 << 18, size=3).astype(np.int32) if rng.rand() < 0.5 else np.zeros(3, np.int32)
    return (pts + shift).astype(np.int32)


t_end = time.time() + budget
it = bad = scenes = points = one_tree = 0
while time.time() < t_end:
    K = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 40]))
    k = int(rng.choice([3, 5]))
    clog = int(rng.choice([6, 8, 10, 11]))
    model = models[k]
    clouds = [scene(i) for i in range(K)]
    if sum(len(c) for c in clouds) > 1_500_000:
        continue
    xs = [torch.tensor(c, device=dev) for c in clouds]
    datas, _, b_enc = _encode_batch(xs, model, clog, [1] * K)
    solo = [_encode_to_bytes(x, model, clog, 1)[0] for x in xs]
    outs, _, _, b_dec = _decode_batch(datas, model, dev)
    ok = True
    for q in range(K):
        if datas[q] != solo[q]:
            print(f"MISMATCH iter {it} scene {q}: batched bytes != solo bytes ({len(datas[q])} / {len(solo[q])})", flush=True); ok = False; break
        so, _, _ = _decode_bytes(solo[q], model, dev)
        o = outs[q].cpu().numpy()
        if not np.array_equal(o, so.cpu().numpy()):
            print(f"MISMATCH iter {it} scene {q}: batched decode != solo decode", flush=True); ok = False; break
        if not np.array_equal(lex(o), lex(clouds[q])):
            print(f"MISMATCH iter {it} scene {q}: decoded set != input set", flush=True); ok = False; break
    print(f"iter {it}: K {K} k {k} chunk_log2 {clog} points {[len(c) for c in clouds][:8]}{'...' if K > 8 else ''} one tree enc/dec {b_enc}/{b_dec} {'ok' if ok else 'BAD'}", flush=True)
    it += 1; scenes += K; points += sum(len(c) for c in clouds); one_tree += int(b_enc and b_dec)
    if not ok:
        bad += 1
        np.savez(os.path.join(ROOT, "gpurun_out", f"batch_soak_bad_{seed}_{it}.npz"), **{f"c{q}": c for q, c in enumerate(clouds)}, k=k, clog=clog)
        break
print(f"batch soak: {it} batches, {scenes} scenes, {points} points, {one_tree} batches as one tree (both directions), {bad} bad")
sys.exit(1 if bad else 0)
