#!/usr/bin/env python3
"""Developer probe: S scenes in flight on ONE GPU -- S host threads, each with its own gpcc context, HIP stream and
1 M-point scene, looping encode + decode.  Prints whole-GPU Mpoints/s for S = 1, 2, 3 (the per-scene latency grows; the
idle tails of the conv launches, the range decoder's serial stretches and the small levels of one scene are filled by the
other).  Usage: tools/throughput_probe.py [points] [steps]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gauspcc_amd import _lib, runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
L = _lib.lib()
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)


def worker(i, x, ctx, stream, barrier, out):
    def step():
        pb, nb, st = C.c_void_p(), C.c_int64(), _lib.Stats()
        sp = C.c_void_p(stream.cuda_stream)
        _lib.check(L.gpcc_encode(ctx, model.handle, x.data_ptr(), x.shape[0], 10, runtime.f16_bits(1), C.byref(pb), C.byref(nb), C.byref(st), sp))
        px, nn, pq, st2 = C.c_void_p(), C.c_int64(), C.c_uint16(), _lib.Stats()
        if os.environ.get("PROBE_COPY"):   # through a Python bytes object (pageable memory) instead of the library's pinned buffer
            data = C.string_at(pb, nb.value)
            _lib.check(L.gpcc_decode(ctx, model.handle, C.cast(C.c_char_p(data), C.c_void_p), len(data), C.byref(px), C.byref(nn), C.byref(pq), C.byref(st2), sp))
        else:
            _lib.check(L.gpcc_decode(ctx, model.handle, pb, nb.value, C.byref(px), C.byref(nn), C.byref(pq), C.byref(st2), sp))
        assert nn.value == x.shape[0]
    step()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    stream.synchronize()
    out[i] = time.perf_counter() - t0


if os.environ.get("PROBE_MAIN_FIRST"):   # what bench.py does before its in-flight pass: a context on the default stream
    from gauspcc_amd.pcc_utils import _decode_bytes, _encode_view
    x0 = torch.tensor(synthetic_cloud(n, seed=1234), device=dev)
    for _ in range(2):
        d0, _ = _encode_view(x0, model, 11, 1)
        _decode_bytes(d0, model, dev)
    torch.cuda.synchronize()

for S in (1, 2, 3):
    xs = [torch.tensor(synthetic_cloud(n, seed=1234 + i), device=dev) for i in range(S)]
    ctxs = []
    for i in range(S):
        h = C.c_void_p()
        _lib.check(L.gpcc_ctx_create(0, C.byref(h)))
        ctxs.append(h)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    torch.cuda.synchronize()
    barrier = threading.Barrier(S)
    out = [0.0] * S
    th = [threading.Thread(target=worker, args=(i, xs[i], ctxs[i], streams[i], barrier, out)) for i in range(S)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = max(out)
    print(f"{S} scene(s) in flight: {S * steps * n / el / 1e6:6.2f} Mpoints/s  ({1e3 * el / steps:.1f} ms per scene step)")
    for h in ctxs:
        L.gpcc_ctx_destroy(h)
