#!/usr/bin/env python3
"""Developer probe: gpcc_conv3d on a dense s x s x s cube (every interior node has all 125 neighbours, as on the coarse
octree levels) -- run under `rocprofv3 --kernel-trace --stats` to time k_sparse_conv_coop.  Usage: tools/coop_probe.py [side]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import gpu_helpers as gh

s = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = np.stack(np.meshgrid(np.arange(s), np.arange(s), np.arange(s), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
xyz = g[gh.sort_zyx(g)]
rng = np.random.RandomState(1)
f = rng.randn(len(xyz), 32).astype(np.float32)
w = (rng.randn(125, 32, 32) * 0.1).astype(np.float32)
for _ in range(20):
    out, pairs = gh.conv3d(xyz, f, w, 5, relu=True)
print(len(xyz), "points", pairs, "pairs")
