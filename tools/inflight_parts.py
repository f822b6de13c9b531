#!/usr/bin/env python3
"""Race hunt, part by part: one thread loops full encode + decode of a 1 M-point scene (the load), the other repeats ONE
stage entry point on fixed inputs and compares every result with its first.  Usage: inflight_parts.py [iterations]"""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

from gauspcc_amd import _lib, runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
from tests import gpu_helpers as gh

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
L = _lib.lib()
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
stop = threading.Event()
load_bad = []


def load():
    x = torch.tensor(synthetic_cloud(1_000_000, seed=77), device=dev)
    h = C.c_void_p()
    _lib.check(L.gpcc_ctx_create(0, C.byref(h)))
    s = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(s.cuda_stream)
    k = 0
    while not stop.is_set():
        pb, nb, st = C.c_void_p(), C.c_int64(), _lib.Stats()
        rc = L.gpcc_encode(h, model.handle, x.data_ptr(), x.shape[0], 11, runtime.f16_bits(1), C.byref(pb), C.byref(nb), C.byref(st), sp)
        px, nn, pq, st2 = C.c_void_p(), C.c_int64(), C.c_uint16(), _lib.Stats()
        if rc == 0:
            rc = L.gpcc_decode(h, model.handle, pb, nb.value, C.byref(px), C.byref(nn), C.byref(pq), C.byref(st2), sp)
        if rc:
            load_bad.append((k, L.gpcc_last_error().decode(errors="replace")))
        k += 1
    s.synchronize()
    L.gpcc_ctx_destroy(h)
    print(f"load thread: {k} steps, {len(load_bad)} bad", load_bad[:3])


def parts():
    rng = np.random.RandomState(1)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        # range coder, the three row widths, 1 M symbols
        for lp in (3, 5, 17):
            n = 1_000_000
            p = rng.dirichlet(np.ones(lp - 1) * 2.0, size=n).astype(np.float32)
            cdf = np.concatenate([np.zeros((n, 1), np.float32), np.cumsum(p, 1)], 1).clip(0, 1)
            ci = np.minimum(np.rint(cdf * (65536 - (lp - 1))).astype(np.int64) + np.arange(lp)[None, :], 65535).astype(np.uint16)
            ci[:, -1] = 0
            sym = (rng.rand(n, 1) > np.cumsum(p, 1)).sum(1).clip(0, lp - 2).astype(np.uint8)
            ref = gh.rc_encode(ci, sym, 11)
            bad_e = bad_d = 0
            for _ in range(iters):
                d = gh.rc_encode(ci, sym, 11)
                bad_e += d != ref
                bad_d += not np.array_equal(gh.rc_decode(ci, ref, 11), sym)
            print(f"rc lp {lp}: encode bad {bad_e}, decode bad {bad_d} of {iters}", flush=True)
        # one convolution, 300 k points (the 255-row pair loop) and 6 k points (cooperative kernel)
        for npts in (300_000, 6_000):
            pts = synthetic_cloud(npts, seed=5)
            xyz = pts[gh.sort_zyx(pts)]
            f = rng.randn(len(xyz), 32).astype(np.float32)
            w = (rng.randn(125, 32, 32) * 0.1).astype(np.float32)
            ref, _ = gh.conv3d(xyz, f, w, 5, relu=True)
            bad = sum(not np.array_equal(gh.conv3d(xyz, f, w, 5, relu=True)[0], ref) for _ in range(iters))
            print(f"conv3d {npts}: bad {bad} of {iters}", flush=True)
        # heads
        for m in (2, 4, 16):
            x = (rng.randn(500_000, 32) * 2).astype(np.float32)
            w1 = rng.rand(32, 32).astype(np.float32) - 0.5; b1 = rng.rand(32).astype(np.float32) - 0.5
            w2 = rng.rand(m, 32).astype(np.float32) - 0.5; b2 = rng.rand(m).astype(np.float32) - 0.5
            _, ref = gh.head_cdf(x, w1, b1, w2, b2)
            bad = sum(not np.array_equal(gh.head_cdf(x, w1, b1, w2, b2)[1], ref) for _ in range(iters))
            print(f"head m {m}: bad {bad} of {iters}", flush=True)
        # octree build
        pts = synthetic_cloud(300_000, seed=9)
        ref = gh.build_octree(pts)
        bad = 0
        for _ in range(iters):
            t = gh.build_octree(pts)
            bad += not (len(t) == len(ref) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(t, ref)))
        print(f"octree: bad {bad} of {iters}", flush=True)


tl = threading.Thread(target=load)
tl.start()
try:
    parts()
finally:
    stop.set()
    tl.join()
