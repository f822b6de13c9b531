"""Mirror of the torchac 0.9.3 interface the reference uses (requirements.txt:6): geometry streams
in pcc_utils.py:174-177, 322-366 and the TC-GS / CAT-3DGS attribute codecs
(TC-GS/utils/encodings.py:38-174, CAT-3DGS/utils/encodings.py:39-175):

    encode_float_cdf(cdf_float, sym, needs_normalization=True, check_input_bounds=False) -> bytes
    decode_float_cdf(cdf_float, byte_stream, needs_normalization=True) -> int16 tensor
    encode_int16_normalized_cdf(cdf_int, sym) -> bytes
    decode_int16_normalized_cdf(cdf_int, byte_stream) -> int16 tensor

Byte-compatible with torchac (one range-coder stream for the whole tensor), computed on the MI355X.
torchac takes CPU tensors; here CPU tensors are moved to the current GPU and CUDA tensors are used in
place.  A single stream decodes on a single lane, so for large tensors prefer the chunked
gauspcc_amd.arithmetic interface (the reference's own CUDA coder format).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, runtime


def _dev(t):
    return t if t.is_cuda else t.to(torch.device("cuda", torch.cuda.current_device()))


def _flatten(cdf, sym=None):
    lp = cdf.shape[-1]
    cdf2 = cdf.reshape(-1, lp).contiguous()
    if sym is not None:
        if tuple(sym.shape) != tuple(cdf.shape[:-1]):
            raise ValueError(f"Invalid shapes of cdf={tuple(cdf.shape)}, sym={tuple(sym.shape)}! The first m elements of cdf.shape must be equal to sym.shape")
        if sym.dtype != torch.int16:
            raise ValueError(f"sym must be int16, got {sym.dtype}")
        sym = sym.reshape(-1).contiguous()
    return cdf2, sym, lp


def _encode(cdf, sym, is_u16):
    cdf, sym = _dev(cdf), _dev(sym)
    n = sym.numel()
    if n == 0:
        return b""
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    fn = _lib.lib().gsac_encode_u16 if is_u16 else _lib.lib().gsac_encode
    _lib.check(fn(runtime.context(cdf.device), sym.data_ptr(), cdf.data_ptr(), n, n, cdf.shape[1], C.byref(pb), C.byref(nb), C.byref(pc), C.byref(nc),
                  runtime.stream_ptr(cdf.device)))
    return C.string_at(pb, nb.value)


def _decode(cdf, byte_stream, is_u16, out_shape, out_device):
    cdf = _dev(cdf)
    n = cdf.shape[0]
    out = torch.zeros(n, dtype=torch.int16, device=cdf.device)
    if n:
        data = np.frombuffer(byte_stream, dtype=np.uint8)
        cnt = np.array([data.size], dtype=np.int32)
        buf = np.ascontiguousarray(data) if data.size else np.zeros(1, np.uint8)
        fn = _lib.lib().gsac_decode_u16 if is_u16 else _lib.lib().gsac_decode
        _lib.check(fn(runtime.context(cdf.device), cdf.data_ptr(), buf.ctypes.data, data.size, cnt.ctypes.data, n, n, cdf.shape[1], out.data_ptr(),
                      runtime.stream_ptr(cdf.device)))
    return out.reshape(out_shape).to(out_device)


def encode_float_cdf(cdf_float, sym, needs_normalization=True, check_input_bounds=False):
    if check_input_bounds:
        if cdf_float.min() < 0:
            raise ValueError(f"cdf_float.min() == {cdf_float.min()}, should be >=0.!")
        if cdf_float.max() > 1:
            raise ValueError(f"cdf_float.max() == {cdf_float.max()}, should be <=1.!")
        lp = cdf_float.shape[-1]
        if sym.max() >= lp - 1:
            raise ValueError("sym.max() >= Lp - 1!")
    if not needs_normalization:
        raise NotImplementedError("needs_normalization=False is not used by the reference")
    cdf2, sym2, _ = _flatten(cdf_float.to(torch.float32), sym)
    return _encode(cdf2, sym2, False)


def decode_float_cdf(cdf_float, byte_stream, needs_normalization=True):
    if not needs_normalization:
        raise NotImplementedError("needs_normalization=False is not used by the reference")
    cdf2, _, _ = _flatten(cdf_float.to(torch.float32))
    return _decode(cdf2, byte_stream, False, tuple(cdf_float.shape[:-1]), cdf_float.device)


def encode_int16_normalized_cdf(cdf_int, sym):
    if cdf_int.dtype != torch.int16:
        raise ValueError(f"cdf must be int16, got {cdf_int.dtype}")
    cdf2, sym2, _ = _flatten(cdf_int, sym)
    return _encode(cdf2, sym2, True)


def decode_int16_normalized_cdf(cdf_int, byte_stream):
    if cdf_int.dtype != torch.int16:
        raise ValueError(f"cdf must be int16, got {cdf_int.dtype}")
    cdf2, _, _ = _flatten(cdf_int)
    return _decode(cdf2, byte_stream, True, tuple(cdf_int.shape[:-1]), cdf_int.device)
