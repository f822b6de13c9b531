#!/usr/bin/env python3
"""K scenes through one chain of launches (gpcc_encode_batch / gpcc_decode_batch), a few times: the command behind the batch
timelines under profiles/ (run under `rocprofv3 --kernel-trace`, summarised by tools/timeline.py).
    python tools/batch_probe.py K POINTS [ITERS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from gauspcc_amd import runtime  # noqa: E402
from gauspcc_amd.pcc_utils import _decode_batch, _encode_batch  # noqa: E402
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict  # noqa: E402

K, n = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
xs = [torch.tensor(synthetic_cloud(n, seed=1300 + i), device=dev) for i in range(K)]
for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    views, _, b = _encode_batch(xs, model, 11, [1] * K, view=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    blobs = [bytes(v) for v in views]
    t1b = time.perf_counter()
    outs, _, _, db = _decode_batch(blobs, model, dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iter {it}: {K} x {n} points, one tree {b and db}: encode {1e3 * (t1 - t0):.2f} ms, decode {1e3 * (t2 - t1b):.2f} ms, {K * n / ((t1 - t0) + (t2 - t1b)) / 1e6:.2f} Mpoints/s", flush=True)
assert all(o.shape[0] == n for o in outs)
from gauspcc_amd import _lib  # noqa: E402

L = _lib.lib()
L.gpcc_debug_launches(1)
_encode_batch(xs, model, 11, [1] * K, view=True)
ke = int(L.gpcc_debug_launches(1))
_decode_batch(blobs, model, dev)
kd = int(L.gpcc_debug_launches(1))
print(f"kernels per batched encode {ke}, per batched decode {kd}", flush=True)
