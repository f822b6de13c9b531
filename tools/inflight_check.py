#!/usr/bin/env python3
"""Race hunt: S scenes in flight on one GPU (own context, stream, host thread each), every encode's bytes and every decode's
points compared with the scene's first (solo) pass.  Usage: inflight_check.py [scenes] [steps] [points]"""
import ctypes as C
import hashlib
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from gauspcc_amd import _lib, runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dev = torch.device("cuda", 0)
L = _lib.lib()
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
bad = []
lock = threading.Lock()
TRACE = os.environ.get("INFLIGHT_TRACE", "0") != "0"
CHUNK_LOG2 = int(os.environ.get("INFLIGHT_CHUNK_LOG2", "11"))   # 0: the reference layout (one lane per stream: slow decodes)
CAP = int(os.environ.get("INFLIGHT_CAPTURE", "-1"))     # capture the buffers whose tag % 100 is this (29: the 16-ary symbols)


def run(i, x, ctx, stream, ref):
    sp = C.c_void_p(stream.cuda_stream)
    pb, nb, st = C.c_void_p(), C.c_int64(), _lib.Stats()
    rc = L.gpcc_encode(ctx, model.handle, x.data_ptr(), x.shape[0], CHUNK_LOG2, runtime.f16_bits(1), C.byref(pb), C.byref(nb), C.byref(st), sp)
    if rc:
        return f"encode rc {rc}: {L.gpcc_last_error().decode(errors='replace')}"
    h = hashlib.sha1(C.string_at(pb, nb.value)).hexdigest()
    etr = None
    if TRACE:
        tags = (C.c_int * 4096)(); sums = (C.c_ulonglong * 4096)()
        k = L.gpcc_debug_trace_get(ctx, tags, sums, 4096)
        etr = [(tags[j], sums[j]) for j in range(k)]
        if ref is not None and h != ref[0]:
            rt = ref[3]
            return "encode bytes differ; trace: " + " ".join(f"{a[0]}{'=' if a == b else '!'}" for a, b in zip(etr, rt))
    px, nn, pq, st2 = C.c_void_p(), C.c_int64(), C.c_uint16(), _lib.Stats()
    rc = L.gpcc_decode(ctx, model.handle, pb, nb.value, C.byref(px), C.byref(nn), C.byref(pq), C.byref(st2), sp)
    tr = None
    if TRACE:
        tags = (C.c_int * 4096)(); sums = (C.c_ulonglong * 4096)()
        k = L.gpcc_debug_trace_get(ctx, tags, sums, 4096)
        tr = [(tags[j], sums[j]) for j in range(k)]
    if rc:
        msg = f"decode rc {rc}: {L.gpcc_last_error().decode(errors='replace')} (bytes {'same' if ref and h == ref[0] else 'DIFFER'})"
        if TRACE and ref and h == ref[0]:
            rt = ref[2]
            first = next((j for j in range(min(len(tr), len(rt))) if tr[j] != rt[j]), None)
            msg += f"; trace {len(tr)} / {len(rt)} marks, first difference at {first}: {tr[first] if first is not None else None} vs {rt[first] if first is not None else None}"
            if first is not None:
                msg += " | differing tags: " + " ".join(str(tr[j][0]) for j in range(min(len(tr), len(rt))) if tr[j] != rt[j])[:120]
                tag = tr[first][0]
                if tag % 100 == CAP and tag in ref[4]:
                    import numpy as np
                    want = ref[4][tag]
                    got = np.empty_like(want)
                    L.gpcc_debug_capture_get(ctx, tag, got.ctypes.data, got.size)
                    idx = np.nonzero(got != want)[0]
                    n = want.size
                    c = 0
                    while (1 << c) < (n + 127) // 128: c += 1
                    c = min(max(c, 7), 11); S = 1 << (c - 1)
                    lanes = sorted(set((idx // S).tolist()))
                    msg += f" | symbols {n}, lane size {S}: {idx.size} differ, lanes {lanes[:20]}, first at lane {idx[0] // S} offset {idx[0] % S}" if idx.size else " | captured symbols equal?!"
                    for L0 in lanes[:4]:
                        ii = idx[idx // S == L0]
                        msg += f" | lane {L0}: offsets {int(ii[0] % S)}..{int(ii[-1] % S)} ({ii.size}), got {got[ii[0]:ii[0] + 12].tolist()} want {want[ii[0]:ii[0] + 12].tolist()}"
        return msg
    with torch.cuda.stream(stream):
        out = torch.empty((nn.value, 3), dtype=torch.int32, device=dev)
        _lib.check(L.gpcc_memcpy_d2d(ctx, C.c_void_p(out.data_ptr()), px, 12 * nn.value, sp))
        s = (out.to(torch.int64) * torch.tensor([1, 1 << 21, 1 << 42], device=dev)).sum(1).sort().values
    stream.synchronize()
    if ref is None:
        caps = {}
        if TRACE and CAP >= 0:
            import numpy as np
            for tag, _ in tr:
                if tag % 100 == CAP:
                    nbytes = L.gpcc_debug_capture_get(ctx, tag, None, 0)
                    if nbytes > 0:
                        a = np.empty(nbytes, np.uint8)
                        L.gpcc_debug_capture_get(ctx, tag, a.ctypes.data, nbytes)
                        caps[tag] = a
        return (h, s, tr, etr, caps)
    msg = []
    if h != ref[0]:
        msg.append("encode bytes differ")
    if s.shape != ref[1].shape or not torch.equal(s, ref[1]):
        msg.append("decoded points differ")
    return ", ".join(msg) if msg else None


def worker(i, x, ctx, stream, barrier):
    ref = run(i, x, ctx, stream, None)
    if not isinstance(ref, tuple):   # the solo pass itself failed (it runs beside the other scenes' solo passes)
        with lock:
            bad.append((i, -1, ref))
            print("  scene %d solo pass: %s" % (i, ref[:600]), flush=True)
        barrier.abort()
        return
    try:
        barrier.wait()
    except threading.BrokenBarrierError:
        return
    for k in range(steps):
        r = run(i, x, ctx, stream, ref)
        if r:
            with lock:
                bad.append((i, k, r))
                print("  scene %d step %d: %s" % (i, k, r[:600]), flush=True)


xs = [torch.tensor(synthetic_cloud(n, seed=1234 + 100 * i), device=dev) for i in range(S)]
ctxs = []
for i in range(S):
    h = C.c_void_p()
    _lib.check(L.gpcc_ctx_create(0, C.byref(h)))
    ctxs.append(h)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
if TRACE:
    for h in ctxs:
        _lib.check(L.gpcc_debug_trace_enable(h, 1))
        _lib.check(L.gpcc_debug_capture(h, CAP))
torch.cuda.synchronize()
barrier = threading.Barrier(S)
th = [threading.Thread(target=worker, args=(i, xs[i], ctxs[i], streams[i], barrier)) for i in range(S)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"{S} scenes x {steps} steps: {len(bad)} bad", flush=True)
