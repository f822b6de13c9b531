#!/usr/bin/env python3
"""Developer probe: time gpcc_encode only (conv roofline from the in-library HIP events).  Used for kernel
experiments whose numerics are deliberately broken (the decoder would reject the stream)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gauspcc_amd import _lib, runtime
from gauspcc_amd.pcc_utils import _encode_to_bytes
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
x = torch.tensor(synthetic_cloud(n, seed=1234), device=dev)
ctx = runtime.context(dev)
L = _lib.lib()
_encode_to_bytes(x, model, 11, 1)
_lib.check(L.gpcc_profile_enable(ctx, 1))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    data, st = _encode_to_bytes(x, model, 11, 1)
torch.cuda.synchronize()
t1 = time.perf_counter()
prof = _lib.Profile()
_lib.check(L.gpcc_profile_get(ctx, C.byref(prof)))
fl = 2048.0 * prof.conv_pair_jobs
print(f"enc_ms {1e3 * (t1 - t0) / steps:.2f}  conv_ms/step {prof.conv_ms / steps:.2f}  conv TFLOP/s {fl / (prof.conv_ms * 1e-3) / 1e12:.1f}  bytes {len(data)}")
