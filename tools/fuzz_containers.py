#!/usr/bin/env python3
"""Corrupted containers at scale: payload runs of a 1 M-point container randomised, truncations, header bytes -- the decoder
must return an error or some cloud, never fault (the small clouds of tests/test_gpu_parity.py never left mapped memory)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gauspcc_amd import runtime
from gauspcc_amd._lib import GpccError
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
from tests import gpu_helpers as gh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # corruptions 0 .. skip - 1 are drawn but not decoded
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
pts = synthetic_cloud(n, seed=11)
clog = int(os.environ.get("FUZZ_CHUNK_LOG2", "11"))   # 0: the reference layout (its coder runs on the host: csrc/hostcoder.hpp)
data, _ = gh.encode(model, pts, clog)
rng = np.random.RandomState(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
out = {"decoded": 0, "error": 0}
for it in range(iters):
    b = bytearray(data)
    mode = it % 4
    if mode == 0:
        i = rng.randint(len(b) // 50, len(b) - 64)          # anywhere behind the first levels
        b[i:i + 32] = rng.randint(0, 256, 32).astype(np.uint8).tobytes()
    elif mode == 1:
        i = rng.randint(200, len(b) // 100)                 # the small levels: everything below them is garbage
        b[i:i + 8] = rng.randint(0, 256, 8).astype(np.uint8).tobytes()
    elif mode == 2:
        b = b[: rng.randint(len(b) // 4, len(b))]
    else:
        b[rng.randint(8, 80)] = rng.randint(256)
    if it < skip:
        continue
    try:
        gh.decode(model, bytes(b))
        out["decoded"] += 1
    except GpccError as e:
        out["error"] += 1
    print(it, mode, out, flush=True)
dec, _, _ = gh.decode(model, data)
assert dec.shape == pts.shape
print("fuzz done", out)
