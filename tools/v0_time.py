#!/usr/bin/env python3
"""Developer probe: encode / decode times of the REFERENCE-LAYOUT container (chunk_log2 = 0: the coder on the host, csrc/hostcoder.hpp) at a given size.
    python tools/v0_time.py [points] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gauspcc_amd import runtime  # noqa: E402
from gauspcc_amd.pcc_utils import _decode_bytes, _encode_view  # noqa: E402
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
x = torch.tensor(synthetic_cloud(n, seed=1234), device=dev)
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v, st = _encode_view(x, model, 0, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    blob = bytes(v)
    t1b = time.perf_counter()
    out, _, _ = _decode_bytes(blob, model, dev)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"reference layout, {n} points: encode {1e3 * (t1 - t0):.1f} ms, decode {1e3 * (t2 - t1b):.1f} ms, {n / ((t1 - t0) + (t2 - t1b)) / 1e6:.2f} Mpoints/s, {len(blob)} bytes", flush=True)
