"""Mirror of the torchac 0.9.3 interface the reference uses (requirements.txt:6): geometry streams
in pcc_utils.py:174-177, 322-366 and the TC-GS / CAT-3DGS attribute codecs
(TC-GS/utils/encodings.py:38-174, CAT-3DGS/utils/encodings.py:39-175):

    encode_float_cdf(cdf_float, sym, needs_normalization=True, check_input_bounds=False) -> bytes
    decode_float_cdf(cdf_float, byte_stream, needs_normalization=True) -> int16 tensor
    encode_int16_normalized_cdf(cdf_int, sym) -> bytes
    decode_int16_normalized_cdf(cdf_int, byte_stream) -> int16 tensor

Byte-compatible with torchac: ONE range-coder stream for the whole tensor.  That format is a single dependent chain, so
the coder itself runs where such a chain runs fastest -- on a host core, in libgauspcc's plain C++ twin of the device lane
loop (csrc/hostcoder.hip: gsac_host_encode_u16 / gsac_host_decode_u16; round 3 ran it on one GPU lane at 4.8 / 2.2 Msymbols/s,
slower than torchac's own CPU loop).  What the GPU does is the part that is data-parallel: the float CDF table is turned into
torchac's int16 rows with torch ops on the device the table lives on (`_convert_to_int_and_normalize`, torchac.py of the
package = kit/op.py:67-79) and crosses PCIe once, as 2 bytes per entry instead of 4.  torchac takes CPU tensors; CUDA tensors
work too.  Callers that can choose their format get parallel decode from the chunked device coders (gauspcc_amd.arithmetic,
the reference's own CUDA coder format).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

PRECISION = 16


def _flatten(cdf, sym=None):
    lp = cdf.shape[-1]
    cdf2 = cdf.reshape(-1, lp)
    if sym is not None:
        if tuple(sym.shape) != tuple(cdf.shape[:-1]):
            raise ValueError(f"Invalid shapes of cdf={tuple(cdf.shape)}, sym={tuple(sym.shape)}! The first m elements of cdf.shape must be equal to sym.shape")
        if sym.dtype != torch.int16:
            raise ValueError(f"sym must be int16, got {sym.dtype}")
        sym = sym.reshape(-1)
    return cdf2, sym, lp


def _to_int_rows(cdf_float):
    """torchac's _convert_to_int_and_normalize(cdf_float, needs_normalization=True) on the tensor's own device:
    round(cdf * (2^16 - (Lp - 1))) + arange(Lp), kept as the int16 bit pattern.  Through int32 (exact for values up to 2^16):
    a float -> int16 conversion of 32768 .. 65536 is not the same on every backend, the int32 -> int16 truncation is."""
    lp = cdf_float.shape[-1]
    # in the tensor's OWN dtype, as torchac multiplies and rounds it (a float64 table rounds differently from its float32 image:
    # ADVICE round 4); half-precision tables go through float32, the only case torch's CPU backend cannot round natively everywhere
    own = cdf_float if cdf_float.dtype in (torch.float32, torch.float64) else cdf_float.to(torch.float32)
    scaled = own.mul(float(2 ** PRECISION - (lp - 1))).round().to(torch.int32)
    return (scaled + torch.arange(lp, dtype=torch.int32, device=cdf_float.device)).to(torch.int16)


def _host(t):
    """contiguous numpy view of a tensor's data on the host (a CUDA tensor is copied once)"""
    return np.ascontiguousarray(t.detach().cpu().numpy())


def _rows(cdf2):
    """(host array, float?) the coder reads: a float table that already lives on the host goes in as it is (the coder integerises
    row by row: gsac_host_*_f32); a float table on a GPU is integerised there and crosses PCIe as int16; int16 rows as they are"""
    if cdf2.dtype == torch.int16:
        return _host(cdf2), False
    if cdf2.device.type == "cpu" and cdf2.dtype == torch.float32:
        return _host(cdf2), True
    return _host(_to_int_rows(cdf2)), False


def _encode(cdf2, sym):
    n = sym.numel()
    if n == 0:
        return b""
    rows, is_float = _rows(cdf2)
    syms = _host(sym)
    cap = 4 * n + 64
    out = np.empty(cap, dtype=np.uint8)
    nb = C.c_int64()
    fn = _lib.lib().gsac_host_encode_f32 if is_float else _lib.lib().gsac_host_encode_u16
    _lib.check(fn(syms.ctypes.data, rows.ctypes.data, n, rows.shape[1], out.ctypes.data, cap, C.byref(nb)))
    return out[: nb.value].tobytes()


def _decode(cdf2, byte_stream, out_shape, out_device):
    rows, is_float = _rows(cdf2)
    n = rows.shape[0]
    out = np.zeros(n, dtype=np.int16)
    if n:
        data = np.frombuffer(byte_stream, dtype=np.uint8)
        buf = np.ascontiguousarray(data) if data.size else np.zeros(1, np.uint8)
        fn = _lib.lib().gsac_host_decode_f32 if is_float else _lib.lib().gsac_host_decode_u16
        _lib.check(fn(rows.ctypes.data, buf.ctypes.data, data.size, n, rows.shape[1], out.ctypes.data))
    return torch.from_numpy(out).reshape(out_shape).to(out_device)


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
    cdf2, sym2, _ = _flatten(cdf_float, sym)
    return _encode(cdf2, sym2)


def decode_float_cdf(cdf_float, byte_stream, needs_normalization=True):
    if not needs_normalization:
        raise NotImplementedError("needs_normalization=False is not used by the reference")
    cdf2, _, _ = _flatten(cdf_float)
    return _decode(cdf2, byte_stream, tuple(cdf_float.shape[:-1]), cdf_float.device)


def encode_int16_normalized_cdf(cdf_int, sym):
    if cdf_int.dtype != torch.int16:
        raise ValueError(f"cdf must be int16, got {cdf_int.dtype}")
    cdf2, sym2, _ = _flatten(cdf_int, sym)
    return _encode(cdf2, sym2)


def decode_int16_normalized_cdf(cdf_int, byte_stream):
    if cdf_int.dtype != torch.int16:
        raise ValueError(f"cdf must be int16, got {cdf_int.dtype}")
    cdf2, _, _ = _flatten(cdf_int)
    return _decode(cdf2, byte_stream, tuple(cdf_int.shape[:-1]), cdf_int.device)
