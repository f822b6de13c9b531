#!/usr/bin/env python3
"""Developer probe: gshac_mlp2 (mlp_grid shape 96-100-175) at a few row counts: fixed cost of a launch (weights into LDS) against the per-tile rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gauspcc_amd import hac_codec
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(2)
w1 = (torch.randn(100, 96, generator=g) / 10).to(dev); b1 = torch.zeros(100, device=dev)
w2 = (torch.randn(175, 100, generator=g) / 10).to(dev); b2 = torch.zeros(175, device=dev)
for rows in (16, 16384, 65536, 262144, 1000000, 4000000):
    x = torch.randn(rows, 96, generator=g).to(dev)
    hac_codec.mlp2(x, w1, b1, w2, b2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        hac_codec.mlp2(x, w1, b1, w2, b2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"rows {rows:8d}: {1e6 * dt:8.1f} us  {rows * 2.0 * (96 * 100 + 100 * 175) / dt / 1e12:6.2f} TFLOP/s")
