#!/usr/bin/env python3
"""Developer probe: one encode + decode with GAUSPCC_HOST_TRACE=1 (host-side phase times on stderr), plus the time the
Python wrapper spends around the two C calls.  Usage: tools/host_trace.py [points] [profile 0/1]"""
import os
import sys
import time

os.environ["GAUSPCC_HOST_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gauspcc_amd import _lib, runtime
from gauspcc_amd.pcc_utils import _decode_bytes, _encode_to_bytes
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
x = torch.tensor(synthetic_cloud(n, seed=1234), device=dev)
ctx = runtime.context(dev)
os.environ["GAUSPCC_HOST_TRACE"] = "0"
for _ in range(2):
    data, _ = _encode_to_bytes(x, model, 11, 1)
    _decode_bytes(data, model, dev)
_lib.check(_lib.lib().gpcc_profile_enable(ctx, prof))
os.environ["GAUSPCC_HOST_TRACE"] = "1"
torch.cuda.synchronize()
for rep in range(2):
    sys.stderr.write(f"---- step {rep}\n")
    t0 = time.perf_counter()
    data, st = _encode_to_bytes(x, model, 11, 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec, _, st2 = _decode_bytes(data, model, dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sys.stderr.write(f"[py] encode {1e3 * (t1 - t0):.3f} ms (library {st.device_ms:.3f})   decode {1e3 * (t2 - t1):.3f} ms (library {st2.device_ms:.3f})\n")
