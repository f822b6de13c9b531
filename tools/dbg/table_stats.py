"""How many bytes do the chunk tables of a version-3 container take, and what would delta coding save? (CPU, oracle)"""
import sys, time, struct
import numpy as np
sys.path.insert(0, "/root/repo")
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
from gauspcc_amd.model import tensor_table
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = 5
sd = synthetic_state_dict(32, k)
om = orc.Model(tensor_table(sd, 32, k), 32, k)
orc.set_threads(8)
x = synthetic_cloud(n, seed=1234)
t0 = time.time()
data = orc.encode(om, x, chunk_log2=11)
v0 = orc.encode(om, x, chunk_log2=0)
print("encode", time.time() - t0, "s", len(data), len(v0), (len(data) - len(v0)) / (4.0 * n / 8))
b = np.frombuffer(data, np.uint8)
assert b[0] == 255 and b[1] == 255 and b[2] == 3
clog, L = int(b[3]), int(b[6])
off = 8
ns = np.frombuffer(data, "<u4", L, off); off += 4 * L
N = struct.unpack_from("<I", data, off)[0]; off += 4
bn = struct.unpack_from("<I", data, off)[0]; off += 4 + 13 * bn
nst = struct.unpack_from("<H", data, off)[0]; off += 2
print("levels", L, list(ns), "N", N, "streams", nst)
def lclog(n):
    want = (n + 127) // 128; c = 0
    while (1 << c) < want: c += 1
    return min(max(c, 7), clog)
def vsize(v):
    k = 1
    while v >= 128: v >>= 7; k += 1
    return k
tot_tab = tot_delta = tot_delta_mean = nchunks = 0
for st in range(nst):
    ln = struct.unpack_from("<I", data, off)[0]; off += 4
    n = int(ns[st // 4 + 1]); c = lclog(n); S = 1 << (c - 1); nl = -(-n // S); nch = (nl + 1) // 2
    pos = off; cb = []
    for _ in range(nch):
        v = 0; k = 0
        while True:
            by = data[pos]; pos += 1; v |= (by & 127) << (7 * k); k += 1
            if not by & 128: break
        cb.append(v)
    tab = pos - off
    zz = lambda d: (d << 1) ^ (d >> 63) if d >= 0 else ((-d) << 1) - 1
    dl = vsize(cb[0]) + sum(vsize(zz(cb[i] - cb[i - 1])) for i in range(1, nch))
    # against the first chunk (full chunks only differ from it by noise)
    dm = vsize(cb[0]) + sum(vsize(zz(cb[i] - cb[0])) for i in range(1, nch))
    tot_tab += tab; tot_delta += dl; tot_delta_mean += dm; nchunks += nch
    if st % 4 == 0 or n > 200000: print(f"stream {st:2d} n {n:7d} chunk 2^{c} chunks {nch:4d} bytes {ln:7d} table {tab:5d} delta-prev {dl:5d} delta-first {dm:5d}  mean {np.mean(cb):.1f} sd {np.std(cb[:-1]) if nch>1 else 0:.1f}")
    off += ln
print("chunks", nchunks, "table bytes", tot_tab, "delta-prev", tot_delta, "delta-first", tot_delta_mean, "overhead now", len(data) - len(v0))
