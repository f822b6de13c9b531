#!/usr/bin/env python3
"""Developer probe: one decode of the 1 M-point bench cloud on a -DFUSED_TIMING build (tools: see HISTORY.md section 4, "small decode
levels, round 4") -- the persistent launches print launch time, time inside grid barriers and time polling, per level.
Usage: GAUSPCC_LIB=gauspcc_amd/variants/libgauspcc_tm512.so python tools/fused_timing.py [points]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gauspcc_amd import runtime
from gauspcc_amd.pcc_utils import _decode_bytes, _encode_to_bytes
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda", 0)
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
x = torch.tensor(synthetic_cloud(n, seed=1234), device=dev)
data, _ = _encode_to_bytes(x, model, 11, 1)
for i in range(3):
    torch.cuda.synchronize()
    print(f"== decode {i}", flush=True)
    _decode_bytes(data, model, dev)
    torch.cuda.synchronize()
