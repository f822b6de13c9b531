"""Thin torch <-> C-ABI glue for the GPU parity tests (stage-level entry points of
include/gauspcc.h).  Everything here calls libgauspcc.so; nothing computes on the CPU."""
import ctypes as C

import numpy as np
import torch

from gauspcc_amd import _lib, runtime


def dev():
    return torch.device("cuda", 0)


def _st():
    return runtime.stream_ptr(dev())


def sort_zyx(xyz: np.ndarray) -> np.ndarray:
    x = torch.tensor(np.ascontiguousarray(xyz, dtype=np.int32), device=dev())
    perm = torch.empty(x.shape[0], dtype=torch.int32, device=dev())
    _lib.check(_lib.lib().gpcc_sort_zyx(runtime.context(dev()), x.data_ptr(), x.shape[0], perm.data_ptr(), _st()))
    return perm.cpu().numpy().astype(np.int64)


def build_octree(xyz: np.ndarray):
    x = torch.tensor(np.ascontiguousarray(xyz, dtype=np.int32), device=dev())
    n = x.shape[0]
    levels = C.c_int32()
    nodes = (C.c_int64 * 24)()
    coords = [np.zeros((n, 3), dtype=np.int32) for _ in range(24)]
    occ = [np.zeros(n, dtype=np.uint8) for _ in range(24)]
    pc = (C.c_void_p * 24)(*[a.ctypes.data for a in coords])
    po = (C.c_void_p * 24)(*[a.ctypes.data for a in occ])
    _lib.check(_lib.lib().gpcc_build_octree(runtime.context(dev()), x.data_ptr(), n, C.byref(levels), nodes, pc, po, n, _st()))
    return [(coords[d][: nodes[d]].copy(), occ[d][: nodes[d]].copy()) for d in range(levels.value)]


def conv3d(xyz_sorted: np.ndarray, feats: np.ndarray, w: np.ndarray, k: int, res=None, relu=False, plan=False):
    x = torch.tensor(np.ascontiguousarray(xyz_sorted, dtype=np.int32), device=dev())
    f = torch.tensor(np.ascontiguousarray(feats, dtype=np.float32), device=dev())
    r = None if res is None else torch.tensor(np.ascontiguousarray(res, dtype=np.float32), device=dev())
    out = torch.empty_like(f)
    w = np.ascontiguousarray(w, dtype=np.float32)
    pairs = C.c_int64()
    _lib.check(_lib.lib().gpcc_conv3d(runtime.context(dev()), x.data_ptr(), x.shape[0], f.shape[1], k, f.data_ptr(), w.ctypes.data,
                                      None if r is None else r.data_ptr(), int(relu) | (2 if plan else 0), out.data_ptr(), C.byref(pairs), _st()))
    return out.cpu().numpy(), pairs.value


def head_cdf(x: np.ndarray, w1, b1, w2, b2):
    m = w2.shape[0]
    xt = torch.tensor(np.ascontiguousarray(x, dtype=np.float32), device=dev())
    prob = torch.empty((xt.shape[0], m), dtype=torch.float32, device=dev())
    cdf = torch.empty((xt.shape[0], m + 1), dtype=torch.int16, device=dev())
    a = [np.ascontiguousarray(t, dtype=np.float32) for t in (w1, b1, w2, b2)]
    _lib.check(_lib.lib().gpcc_head_cdf(runtime.context(dev()), xt.data_ptr(), xt.shape[0], xt.shape[1], m, a[0].ctypes.data, a[1].ctypes.data,
                                        a[2].ctypes.data, a[3].ctypes.data, prob.data_ptr(), cdf.data_ptr(), _st()))
    return prob.cpu().numpy(), cdf.cpu().numpy().view(np.uint16)


def set_version(version: int = 4):
    _lib.check(_lib.lib().gpcc_ctx_set_container_version(runtime.context(dev()), version))


def rc_encode(cdf_u16: np.ndarray, sym: np.ndarray, chunk_log2: int, version: int = 4) -> bytes:
    set_version(version)
    c = torch.tensor(np.ascontiguousarray(cdf_u16).view(np.int16), device=dev())
    s = torch.tensor(np.ascontiguousarray(sym, dtype=np.uint8), device=dev())
    pb, nb = C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().gpcc_rc_encode(runtime.context(dev()), c.data_ptr(), c.shape[1], s.data_ptr(), s.shape[0], chunk_log2,
                                         C.byref(pb), C.byref(nb), _st()))
    return C.string_at(pb, nb.value)


def rc_decode(cdf_u16: np.ndarray, data: bytes, chunk_log2: int, version: int = 4) -> np.ndarray:
    set_version(version)
    c = torch.tensor(np.ascontiguousarray(cdf_u16).view(np.int16), device=dev())
    out = torch.empty(c.shape[0], dtype=torch.uint8, device=dev())
    buf = (C.c_char * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    _lib.check(_lib.lib().gpcc_rc_decode(runtime.context(dev()), c.data_ptr(), c.shape[1], C.cast(buf, C.c_void_p), len(data), c.shape[0],
                                         chunk_log2, out.data_ptr(), _st()))
    return out.cpu().numpy()


def encode(model, xyz: np.ndarray, chunk_log2=11, posq=1, ideal_bits=False, version=None):
    from gauspcc_amd.pcc_utils import _encode_to_bytes

    x = torch.tensor(np.ascontiguousarray(xyz, dtype=np.int32), device=dev())
    return _encode_to_bytes(x, model, chunk_log2, posq, ideal_bits, version)


def decode(model, data: bytes):
    from gauspcc_amd.pcc_utils import _decode_bytes

    out, posq, st = _decode_bytes(data, model, dev())
    return out.cpu().numpy(), posq, st
