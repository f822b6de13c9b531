import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import gpu_helpers as gh
from oracle import oracle as orc
from gauspcc_amd.synth import synthetic_cloud
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pts = synthetic_cloud(n, seed=7)
xyz = pts[gh.sort_zyx(pts)]
rng = np.random.RandomState(1)
f = rng.randn(len(xyz), 32).astype(np.float32)
w = (rng.randn(125, 32, 32) * 0.1).astype(np.float32)
out, pairs = gh.conv3d(xyz, f, w, 5, relu=False)
nb = orc.nbr(xyz, 5)
ref = orc.conv(f, nb, w)
bad = np.nonzero((out != ref).any(axis=1))[0]
print("n", len(xyz), "pairs", pairs, "bad rows", len(bad))
if len(bad):
    print("first bad rows", bad[:20], "blocks(H?)", bad[:20] // 255)
    r = bad[0]
    print("row", r, "max abs diff", np.abs(out[r] - ref[r]).max(), "nbrs", int((nb[r] >= 0).sum()))
    d = np.abs(out - ref).max(axis=1)
    print("diff histogram", np.histogram(d[bad], bins=[0, 1e-6, 1e-4, 1e-2, 1, 1e9])[0])
# dense cube: long runs (full pair steps)
s_ = 46
g = np.stack(np.meshgrid(np.arange(s_), np.arange(s_), np.arange(s_), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
xyz = g[gh.sort_zyx(g)]
f = rng.randn(len(xyz), 32).astype(np.float32)
out, pairs = gh.conv3d(xyz, f, w, 5, relu=False)
nb = orc.nbr(xyz, 5)
ref = orc.conv(f, nb, w)
bad = np.nonzero((out != ref).any(axis=1))[0]
print("cube n", len(xyz), "pairs", pairs, "bad rows", len(bad))
if len(bad):
    print("first bad rows", bad[:30])
    d = np.abs(out - ref).max(axis=1)
    print("diff histogram", np.histogram(d[bad], bins=[0, 1e-6, 1e-4, 1e-2, 1, 1e9])[0])
    H = -(-len(xyz) // 1024)
    print("H", H, "bad rows mod H", np.unique(bad % H)[:40])
