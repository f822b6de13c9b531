#!/usr/bin/env python3
"""Regenerates tests/golden/containers.npz: bitstreams of THIS implementation's oracle for a fixed small cloud and the
seeded synthetic weights (gauspcc_amd.synth), one per container layout.  Unlike the other fixtures these are not derived
from the reference (its coder and sparse-conv packages are absent); they pin the container format and
the normative numerics across rounds -- a change that alters a single byte of a stream fails tests/test_containers.py
until this script is re-run on purpose.  Usage: python tests/golden/make_containers.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gauspcc_amd.model import tensor_table  # noqa: E402
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
pts = synthetic_cloud(700, seed=2024)
out = {"points": pts}
for k in (5, 3):
    m = orc.Model(tensor_table(synthetic_state_dict(32, k), 32, k), 32, k)
    # k*_chunk*: the reference layout (0) and container version 2, which readers must keep reading (round 2's writer);
    # k*_chunk*_v3: version 3, what the encoder writes now (two lanes per byte-counted chunk, LEB128 counts)
    orc.set_container_version(2)
    for cl in (0, 6, 10):
        out[f"k{k}_chunk{cl}"] = np.frombuffer(orc.encode(m, pts, chunk_log2=cl), dtype=np.uint8)
    orc.set_container_version(3)
    for cl in (6, 11):
        out[f"k{k}_chunk{cl}_v3"] = np.frombuffer(orc.encode(m, pts, chunk_log2=cl), dtype=np.uint8)
    # k*_chunk*_v4: version 4 (round 4: the version-3 layout with the carry-propagating coder in the lanes), what the encoder writes now
    orc.set_container_version(4)
    for cl in (6, 11):
        out[f"k{k}_chunk{cl}_v4"] = np.frombuffer(orc.encode(m, pts, chunk_log2=cl), dtype=np.uint8)
# what earlier rounds stored must come out again byte for byte: a new layout is ADDED, the old streams never change
path = os.path.join(ROOT, "tests", "golden", "containers.npz")
if os.path.exists(path):
    old = np.load(path)
    for key in old.files:
        assert key in out and np.array_equal(old[key], out[key]), f"stored fixture {key} would change"
np.savez_compressed(path, **out)
print({k: v.shape for k, v in out.items()})
