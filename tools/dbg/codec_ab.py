#!/usr/bin/env python3
"""Developer probe: encode + decode of the 1 M-point bench cloud, 12 warm steps: ms per call (compare two builds with GAUSPCC_LIB=...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gauspcc_amd import runtime
from gauspcc_amd.pcc_utils import _decode_bytes, _encode_to_bytes
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
dev = torch.device("cuda", 0)
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
x = torch.tensor(synthetic_cloud(1_000_000, seed=1234), device=dev)
for _ in range(4):
    data, _ = _encode_to_bytes(x, model, 11, 1); _decode_bytes(data, model, dev)
te = td = 0.0
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data, _ = _encode_to_bytes(x, model, 11, 1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    _decode_bytes(data, model, dev)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    te += t1 - t0; td += t2 - t1
print(f"{os.environ.get('GAUSPCC_LIB', 'product')}: enc {1e3 * te / 12:.2f} ms dec {1e3 * td / 12:.2f} ms")
