"""ctypes wrapper around oracle/liborc.so plus the numpy restatements of the
pure-python reference helpers.  TEST INFRASTRUCTURE ONLY (see gpcc_oracle.c header):
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by gauspcc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
STAGE_M = (2, 2, 4, 16)


def build(force: bool = False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "gpcc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.orc_last_error.restype = C.c_char_p
        L.orc_set_threads.restype = i32
        L.orc_set_threads.argtypes = [i32]
        L.orc_tree_build.restype = vp
        L.orc_tree_build.argtypes = [vp, i64]
        L.orc_tree_free.argtypes = [vp]
        L.orc_tree_nlevels.argtypes = [vp]
        L.orc_tree_level.restype = i64
        L.orc_tree_level.argtypes = [vp, i32, vp, vp]
        L.orc_nbr.argtypes = [vp, vp, i64, i32, vp]
        L.orc_conv.argtypes = [vp, vp, vp, i64, i32, i32, vp, i32, vp]
        L.orc_head.argtypes = [vp, i64, i32, vp, vp, vp, vp, i32, vp, vp]
        L.orc_cdf_to_int16.argtypes = [vp, i64, i32, vp]
        L.orc_expf_test.restype = C.c_float
        L.orc_expf_test.argtypes = [C.c_float]
        L.orc_rc_encode.restype = i64
        L.orc_rc_encode.argtypes = [vp, i32, vp, i64, vp, i64]
        L.orc_rc_decode.argtypes = [vp, i32, vp, i64, i64, vp]
        L.orc_cp_encode.restype = i64
        L.orc_cp_encode.argtypes = [vp, i32, vp, i64, vp, i64]
        L.orc_cp_decode.argtypes = [vp, i32, vp, i64, i64, vp]
        L.orc_stream_encode.restype = i64
        L.orc_stream_encode.argtypes = [vp, i32, vp, i64, i32, i32, vp, i64]
        L.orc_stream_decode.argtypes = [vp, i32, vp, i64, i64, i32, i32, vp]
        L.orc_chunk_table_put.restype = i64
        L.orc_chunk_table_put.argtypes = [vp, i64, vp]
        L.orc_chunk_table_get.restype = i64
        L.orc_chunk_table_get.argtypes = [vp, i64, i64, vp]
        L.orc_set_container_version.argtypes = [i32]
        L.orc_encode.restype = i64
        L.orc_encode.argtypes = [vp, i32, i32, vp, i64, i32, C.c_uint16, vp, i64]
        L.orc_decode.restype = i64
        L.orc_decode.argtypes = [vp, i32, i32, vp, i64, vp, i64, vp]
        L.orc_trace_enable.argtypes = [i32]
        L.orc_trace_get.restype = i64
        L.orc_trace_get.argtypes = [i32, i32, vp, vp, vp, vp]
        L.orc_ideal_bits.restype = C.c_double
        L.orc_gaussian_cdf.argtypes = [vp, vp, vp, i64, i32, i32, vp]
        L.orc_gaussian_mixed_cdf.argtypes = [vp, vp, vp, i32, vp, i64, i32, i32, vp]
        L.orc_hac_encode.restype = i64
        L.orc_hac_encode.argtypes = [vp, i32, vp, i64, i32, vp, i64, vp]
        L.orc_hac_decode.argtypes = [vp, i32, vp, vp, i64, i32, vp]
        L.orc_grid_forward.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]
        L.orc_mlp2.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
        L.orc_raster_forward.restype = i64
        L.orc_raster_forward.argtypes = [i32, vp, i32, i32, vp, vp, vp, vp, C.c_float, vp, vp, vp, C.c_float, C.c_float, vp, vp]
        _LIB = L
    return _LIB


def set_threads(n: int = 0) -> int:
    """Pin the OpenMP thread count of the oracle's parallel loops (0: leave it); returns the count in force."""
    return int(lib().orc_set_threads(int(n)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _err():
    return lib().orc_last_error().decode()


# ------------------------------------------------------------------ pure-numpy restatements
def raster_order(x: np.ndarray) -> np.ndarray:
    """calculate_morton_order (HAC/utils/pcc_utils.py:12-22): argsort of
    x + y*M + z*M^2 after the per-axis min shift, M = max over all axes + 1.
    kind='stable' (the reference's default introsort gives the same permutation
    whenever points are unique)."""
    x = np.asarray(x)
    assert x.ndim == 2 and x.shape[1] == 3
    x = x - x.min(axis=0, keepdims=True)
    x = x.astype(np.int64)
    m = x.max() + 1
    key = x @ np.power(m, np.arange(3))
    return np.argsort(key, kind="stable")


def sort_cf(coords: np.ndarray, feats: np.ndarray):
    """kit/op.py:17-30: order rows by (batch, z, y, x)."""
    c = np.asarray(coords)
    order = np.lexsort((c[:, 1], c[:, 2], c[:, 3], c[:, 0]))
    return c[order], np.asarray(feats)[order]


def pack_byte_stream_ls(streams) -> bytes:
    """kit/op.py:32-37."""
    out = np.array(len(streams), dtype=np.uint16).tobytes()
    for s in streams:
        out += np.array(len(s), dtype=np.uint32).tobytes() + bytes(s)
    return out


def unpack_byte_stream(stream: bytes):
    """kit/op.py:39-48."""
    n = int(np.frombuffer(stream[:2], dtype=np.uint16)[0])
    cur, out = 2, []
    for _ in range(n):
        ln = int(np.frombuffer(stream[cur:cur + 4], dtype=np.uint32)[0])
        out.append(stream[cur + 4:cur + 4 + ln])
        cur += 4 + ln
    return out


def cdf_to_int16(cdf: np.ndarray) -> np.ndarray:
    """kit/op.py:50-79 (_convert_to_int_and_normalize, needs_normalization=True)."""
    cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    out = np.empty(cdf.shape, dtype=np.int16)
    lib().orc_cdf_to_int16(_p(cdf), cdf.size // cdf.shape[-1], cdf.shape[-1], _p(out))
    return out


def psnr(img1: np.ndarray, img2: np.ndarray) -> np.ndarray:
    """HAC/utils/image_utils.py:17-19: per leading-dim (channel) PSNR, shape (C, 1)."""
    a, b = np.asarray(img1, np.float32), np.asarray(img2, np.float32)
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1, keepdims=True, dtype=np.float64)
    return (20 * np.log10(1.0 / np.sqrt(mse))).astype(np.float32)


# ------------------------------------------------------------------ C oracle
def tree_build(xyz: np.ndarray):
    """[(coords (n,3) int32 raster-sorted, occupancy (n,) uint8), ...] base level first."""
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    L = lib()
    t = L.orc_tree_build(_p(xyz), xyz.shape[0])
    if not t:
        raise ValueError(_err())
    try:
        out = []
        for d in range(L.orc_tree_nlevels(t)):
            px, po = C.c_void_p(), C.c_void_p()
            n = L.orc_tree_level(t, d, C.byref(px), C.byref(po))
            c = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_int32)), shape=(n, 3)).copy()
            o = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint8)), shape=(n,)).copy()
            out.append((c, o))
        return out
    finally:
        L.orc_tree_free(t)


def nbr(xyz_sorted: np.ndarray, k: int) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz_sorted, dtype=np.int32)
    out = np.empty((xyz.shape[0], k ** 3), dtype=np.int32)
    lib().orc_nbr(_p(xyz), None, xyz.shape[0], k, _p(out))
    return out


def conv(x: np.ndarray, nbr_: np.ndarray, w: np.ndarray, res=None, relu=False) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    nb = np.ascontiguousarray(nbr_, dtype=np.int32)
    r = None if res is None else np.ascontiguousarray(res, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_conv(_p(x), _p(nb), _p(w), x.shape[0], x.shape[1], nb.shape[1], _p(r), int(relu), _p(out))
    return out


def head(x, w1, b1, w2, b2):
    x = np.ascontiguousarray(x, dtype=np.float32)
    m = w2.shape[0]
    prob = np.empty((x.shape[0], m), dtype=np.float32)
    cdf = np.empty((x.shape[0], m + 1), dtype=np.uint16)
    a = [np.ascontiguousarray(t, dtype=np.float32) for t in (w1, b1, w2, b2)]
    lib().orc_head(_p(x), x.shape[0], x.shape[1], _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), m, _p(prob), _p(cdf))
    return prob, cdf


def rc_encode(cdf_u16: np.ndarray, sym: np.ndarray) -> bytes:
    cdf = np.ascontiguousarray(cdf_u16).view(np.uint16)
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    cap = sym.size * 4 + 16
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_rc_encode(_p(cdf), cdf.shape[1], _p(sym), sym.size, _p(out), cap)
    assert n <= cap
    return out[:n].tobytes()


def rc_decode(cdf_u16: np.ndarray, data: bytes) -> np.ndarray:
    cdf = np.ascontiguousarray(cdf_u16).view(np.uint16)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(cdf.shape[0], dtype=np.uint8)
    lib().orc_rc_decode(_p(cdf), cdf.shape[1], _p(buf), buf.size, cdf.shape[0], _p(out))
    return out


def cp_encode(cdf_u16: np.ndarray, sym: np.ndarray) -> bytes:
    """One lane of a version-4 container: the carry-propagating range coder (gpcc_oracle.c: cp_encode_core)."""
    cdf = np.ascontiguousarray(cdf_u16).view(np.uint16)
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    cap = sym.size * 4 + 16
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_cp_encode(_p(cdf), cdf.shape[1], _p(sym), sym.size, _p(out), cap)
    assert n <= cap
    return out[:n].tobytes()


def cp_decode(cdf_u16: np.ndarray, data: bytes) -> np.ndarray:
    cdf = np.ascontiguousarray(cdf_u16).view(np.uint16)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(cdf.shape[0], dtype=np.uint8)
    lib().orc_cp_decode(_p(cdf), cdf.shape[1], _p(buf), buf.size, cdf.shape[0], _p(out))
    return out


def stream_encode(cdf_u16: np.ndarray, sym: np.ndarray, chunk_log2: int, version: int = 4) -> bytes:
    """One (level, stage) stream of the container as the codec writes it for a level of len(sym) nodes: the chunk table and
    the chunks (versions 3 / 4: Rice-coded counts, forward + reversed backward lane per chunk; version 4: the carry-propagating
    coder in the lanes); chunk_log2 = 0: the bare bytes of torchac's coder."""
    cdf = np.ascontiguousarray(cdf_u16).view(np.uint16)
    sym = np.ascontiguousarray(sym, dtype=np.uint8)
    cap = sym.size * 4 + 64 + 8 * (sym.size // 32 + 1)
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_stream_encode(_p(cdf), cdf.shape[1], _p(sym), sym.size, chunk_log2, version, _p(out), cap)
    assert 0 <= n <= cap
    return out[:n].tobytes()


def stream_decode(cdf_u16: np.ndarray, data: bytes, chunk_log2: int, version: int = 4) -> np.ndarray:
    cdf = np.ascontiguousarray(cdf_u16).view(np.uint16)
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(cdf.shape[0], dtype=np.uint8)
    if lib().orc_stream_decode(_p(cdf), cdf.shape[1], _p(buf), buf.size, cdf.shape[0], chunk_log2, version, _p(out)):
        raise ValueError("malformed stream")
    return out


def chunk_table(counts) -> bytes:
    """The version-3 chunk table of a stream whose chunks take `counts` bytes."""
    c = np.ascontiguousarray(counts, dtype=np.int64)
    out = np.empty(16 + 6 * c.size, dtype=np.uint8)
    return out[:lib().orc_chunk_table_put(_p(c), c.size, _p(out))].tobytes()


def chunk_table_parse(data: bytes, nch: int):
    """(counts, bytes read) of a version-3 chunk table; ValueError when malformed."""
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(nch, dtype=np.uint32)
    k = lib().orc_chunk_table_get(_p(buf), buf.size, nch, _p(out))
    if not k:
        raise ValueError("malformed chunk table")
    return out, int(k)


def set_container_version(v: int = 4) -> int:
    """Version orc.encode writes for chunk_log2 != 0 (default 4); returns the version in force."""
    return int(lib().orc_set_container_version(int(v)))


class Model:
    """Holds the 39-tensor table alive and exposes it as float**."""

    def __init__(self, tensors, channels=32, kernel_size=5):
        assert len(tensors) == 39
        self.t = [np.ascontiguousarray(a, dtype=np.float32) for a in tensors]
        self.C, self.k = channels, kernel_size
        self.ptrs = (C.c_void_p * 39)(*[a.ctypes.data for a in self.t])


def f16_bits(v) -> int:
    return int(np.array(v, dtype=np.float16).view(np.uint16))


def encode(model: Model, xyz: np.ndarray, chunk_log2: int = 11, posq=1, trace: bool = False) -> bytes:
    xyz = np.ascontiguousarray(xyz, dtype=np.int32)
    n = xyz.shape[0]
    cap = 64 * 1024 + n * 16
    out = np.empty(cap, dtype=np.uint8)
    L = lib()
    L.orc_trace_enable(int(trace))
    nb = L.orc_encode(model.ptrs, model.C, model.k, _p(xyz), n, chunk_log2, f16_bits(posq), _p(out), cap)
    if nb < 0:
        raise ValueError(_err())
    return out[:nb].tobytes()


def ideal_bits() -> float:
    return float(lib().orc_ideal_bits())


def trace():
    """Per coded level of the last encode(trace=True): dict(xyz, sym[4], cdf[4], prob[4])."""
    L = lib()
    out = []
    for d in range(L.orc_trace_levels()):
        lv = {"sym": [], "cdf": [], "prob": []}
        for s, m in enumerate(STAGE_M):
            pc, ps, pp, px = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            n = L.orc_trace_get(d, s, C.byref(pc), C.byref(ps), C.byref(pp), C.byref(px))
            lv["cdf"].append(np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint16)), shape=(n, m + 1)).copy())
            lv["sym"].append(np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint8)), shape=(n,)).copy())
            lv["prob"].append(np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_float)), shape=(n, m)).copy())
            lv["xyz"] = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_int32)), shape=(n, 3)).copy()
        out.append(lv)
    return out


def decode(model: Model, data: bytes, cap_pts: int = None):
    buf = np.frombuffer(data, dtype=np.uint8)
    cap_pts = cap_pts or max(1024, buf.size * 8)
    out = np.empty((cap_pts, 3), dtype=np.int32)
    pq = C.c_uint16(0)
    n = lib().orc_decode(model.ptrs, model.C, model.k, _p(buf), buf.size, _p(out), cap_pts, C.byref(pq))
    if n < 0:
        raise ValueError(_err())
    posq = np.array(pq.value, dtype=np.uint16).view(np.float16)
    return out[:n].copy(), posq


# ------------------------------------------------------------------ HAC attribute coder
def gaussian_cdf(mean, scale, q, min_value: int, max_value: int) -> np.ndarray:
    mean, scale, q = (np.ascontiguousarray(a, dtype=np.float32) for a in (mean, scale, q))
    out = np.empty((mean.size, max_value - min_value + 2), dtype=np.float32)
    lib().orc_gaussian_cdf(_p(mean), _p(scale), _p(q), mean.size, min_value, max_value, _p(out))
    return out


def gaussian_mixed_cdf(means, scales, probs, q, min_value: int, max_value: int) -> np.ndarray:
    """HAC++'s mixture table: clamp(sum_c gaussian_cdf(mean_c, scale_c, q) * prob_c, 0, 1), components in list order."""
    k = len(means)
    arrs = [[np.ascontiguousarray(a, dtype=np.float32) for a in lst] for lst in (means, scales, probs)]
    q = np.ascontiguousarray(q, dtype=np.float32)
    ptrs = [(C.c_void_p * k)(*[a.ctypes.data for a in lst]) for lst in arrs]
    out = np.empty((q.size, max_value - min_value + 2), dtype=np.float32)
    lib().orc_gaussian_mixed_cdf(ptrs[0], ptrs[1], ptrs[2], k, _p(q), q.size, min_value, max_value, _p(out))
    return out


def hac_encode(sym: np.ndarray, cdf: np.ndarray, chunk: int = 10000):
    sym = np.ascontiguousarray(sym, dtype=np.int16)
    cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    n, lp = cdf.shape
    nch = (n + chunk - 1) // chunk
    cap = n * 4 + 16 * nch
    out = np.empty(cap, dtype=np.uint8)
    cnt = np.empty(nch, dtype=np.int32)
    nb = lib().orc_hac_encode(_p(cdf), lp, _p(sym), n, chunk, _p(out), cap, _p(cnt))
    return out[:nb].copy(), cnt


def hac_decode(cdf: np.ndarray, data: np.ndarray, cnt: np.ndarray, chunk: int = 10000) -> np.ndarray:
    cdf = np.ascontiguousarray(cdf, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    out = np.empty(cdf.shape[0], dtype=np.int16)
    lib().orc_hac_decode(_p(cdf), cdf.shape[1], _p(data), _p(cnt), cdf.shape[0], chunk, _p(out))
    return out


def mlp2(x, w1, b1, w2, b2, slope: float = 0.0) -> np.ndarray:
    """mlp_grid (HAC/scene/gaussian_model.py:258-262): Linear - ReLU - Linear in the normative fp32 order; slope != 0: LeakyReLU(slope)
    between the layers (HAC++'s channel-context MLPs, HAC-plus/scene/gaussian_model.py:117-168)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w1, b1, w2, b2 = (np.ascontiguousarray(a, dtype=np.float32) for a in (w1, b1, w2, b2))
    n, din = x.shape
    dh, dout = w1.shape[0], w2.shape[0]
    assert dh <= 1024
    y = np.empty((n, dout), dtype=np.float32)
    lib().orc_mlp2_act(_p(x), _p(w1), _p(b1), _p(w2), _p(b2), C.c_int64(n), din, dh, dout, C.c_float(slope), _p(y))
    return y


def grid_forward(inputs, emb, offsets, resolutions, rb=128, binary_vxl=None, min_level_id=None):
    """_gridencoder.grid_encode_forward restated: returns (n_levels, N, F) float32."""
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    resolutions = np.ascontiguousarray(resolutions, dtype=np.int32)
    n, d = inputs.shape
    f = emb.shape[1]
    nl = resolutions.size
    out = np.empty((nl, n, f), dtype=np.float32)
    bv = None if binary_vxl is None else np.ascontiguousarray(binary_vxl, dtype=np.uint8)
    ml = None if min_level_id is None else np.ascontiguousarray(min_level_id, dtype=np.int32)
    lib().orc_grid_forward(_p(inputs), _p(emb), _p(offsets), _p(resolutions), _p(out), n, d, f, nl, rb, _p(bv), _p(ml))
    return out


def raster_forward(bg, W, H, means, colors, opac, scales, scale_modifier, rots, viewmatrix, projmatrix, tan_fovx, tan_fovy):
    """Rasteriser forward restated.  colors=None -> radii only (visible_filter).
    Returns (image (3,H,W) or None, radii (P,) int32, number of (tile, gaussian) instances)."""
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    means, colors, opac, scales, rots, bg = f(means), f(colors), f(opac), f(scales), f(rots), f(bg)
    V, M = f(viewmatrix).reshape(-1), f(projmatrix).reshape(-1)
    P = means.shape[0]
    radii = np.zeros(P, dtype=np.int32)
    out = None if colors is None else np.zeros((3, H, W), dtype=np.float32)
    n = lib().orc_raster_forward(P, _p(bg), W, H, _p(means), _p(colors), _p(opac), _p(scales), float(scale_modifier), _p(rots), _p(V), _p(M),
                                 float(tan_fovx), float(tan_fovy), _p(out), _p(radii))
    return out, radii, int(n)
