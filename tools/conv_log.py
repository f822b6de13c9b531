#!/usr/bin/env python3
"""Developer probe: per-launch conv times (HIP events inside the library, GAUSPCC_CONV_LOG=1) of one encode + decode,
summed per (level, block class).  Usage: tools/conv_log.py [points]"""
import collections
import os
import re
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time

    import torch

    from gauspcc_amd import _lib, runtime
    from gauspcc_amd.pcc_utils import _decode_bytes, _encode_to_bytes
    from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

    n = int(sys.argv[2])
    dev = torch.device("cuda", 0)
    model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
    x = torch.tensor(synthetic_cloud(n, seed=1234), device=dev)
    ctx = runtime.context(dev)
    for _ in range(2):
        data, _ = _encode_to_bytes(x, model, 11, 1)
        _decode_bytes(data, model, dev)
    _lib.check(_lib.lib().gpcc_profile_enable(ctx, 1))
    torch.cuda.synchronize()
    sys.stderr.write("==encode\n")
    t0 = time.perf_counter()
    data, _ = _encode_to_bytes(x, model, 11, 1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sys.stderr.write("==decode\n")
    _decode_bytes(data, model, dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enc_ms {1e3 * (t1 - t0):.2f} dec_ms {1e3 * (t2 - t1):.2f}")
    sys.exit(0)

n = sys.argv[1] if len(sys.argv) > 1 else "1000000"
env = dict(os.environ, GAUSPCC_CONV_LOG="1")
r = subprocess.run([sys.executable, __file__, "--child", n], env=env, capture_output=True, text=True)
print(r.stdout.strip())
agg = collections.OrderedDict()
phase = None
for line in r.stderr.splitlines():
    if line.startswith("==encode"):
        phase = "enc"
    if line.startswith("==decode"):
        phase = "dec"
    m = re.match(r"\[conv\] level\s+(\d+) n\s+(\d+) R\s+(\d+) H\s+(\d+) blocks\s+(\d+) jobs (\d+)\s+([0-9.]+) us pairs (\d+)", line)
    if m and phase:
        key = (phase, int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5]), int(m[6]))
        a = agg.setdefault(key, [0, 0.0, 0])
        a[0] += 1
        a[1] += float(m[7])
        a[2] = int(m[8])
tot = collections.Counter()
for (ph, lv, nn, R, H, nb, jobs), (c, us, pairs) in agg.items():
    tf = 2048.0 * pairs * jobs * c / (us * 1e-6) / 1e12 if us else 0.0
    print(f"{ph} level {lv:2d} n {nn:8d} R {R:3d} H {H:3d} blocks {nb:6d} jobs {jobs}  launches {c:3d}  avg {us / c:8.1f} us  sum {us / 1e3:7.3f} ms  pairs/row {pairs / max(nn, 1):5.1f}  {tf:6.1f} TFLOP/s")
    tot[ph] += us
print({k: round(v / 1e3, 3) for k, v in tot.items()})
if r.returncode:
    print(r.stderr[-2000:])
