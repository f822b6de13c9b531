"""Mirror of the reference's `arithmetic` extension module
(HAC/submodules/arithmetic.zip!arithmetic/arithmetic.cpp:4-49): same three functions,
same argument order, same tensor types, same byte format -- backed by libgauspcc.so.

    calculate_cdf(mean, scale, Q, min_value, max_value) -> Tensor (n, max-min+2) float32
    arithmetic_encode(sym, cdf, chunk_size, N, Lp)      -> (Tensor uint8, Tensor int32[chunks])
    arithmetic_decode(cdf, bytes, cnt, chunk_size, N, Lp) -> Tensor int16 (N)

plus the two fused calls the `encodings_cuda` mirror uses (no reference counterpart: they replace the
calculate_cdf -> arithmetic_encode / arithmetic_decode pair of encoder_gaussian / decoder_gaussian and never
materialise the (n, max-min+2) float table):

    encode_gaussian(x, mean, scale, Q, chunk_size)                    -> (min, max, Tensor uint8, Tensor int32[chunks])
    decode_gaussian(mean, scale, Q, min, max, bytes, cnt, chunk_size) -> Tensor float32 (n)
    encode_gaussian_slices / decode_gaussian_slices: the same for every slice of an attribute in one call
    encode_gaussian_mixed / decode_gaussian_mixed / calculate_cdf_mixed: HAC++'s K-component mixture
        (HAC-plus/utils/encodings_cuda.py:205-247, 285-317): lower = clamp(sum_c calculate_cdf(mean_c, scale_c, Q) * prob_c, 0, 1)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, runtime



def _owned(ptr, count, dtype):
    """count elements at the library's (pinned, reused) output buffer as an array the caller owns: ONE copy (string_at + frombuffer + copy made two)."""
    if not count:
        return np.empty(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_ubyte * nbytes).from_address(ptr.value), dtype=dtype).copy()

def _chk(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")      # CHECK_CUDA  (include/utils.h:3)
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")          # CHECK_CONTIGUOUS (include/utils.h:4)


def _f32(*ts):
    """float32 contiguous views / copies of the arguments, as a tuple the CALLER keeps alive until the library call has
    returned: `t.float().data_ptr()` on a non-fp32 tensor takes the address of a temporary that the caching allocator hands
    to the next conversion (several arguments would alias one block)."""
    return tuple(t.detach().to(torch.float32).contiguous() for t in ts)


def calculate_cdf(mean, scale, Q, min_value, max_value):
    for t, nm in ((mean, "mean"), (scale, "scale"), (Q, "Q")):
        _chk(t, nm)
    mn, mx = int(min_value), int(max_value)
    n = mean.shape[0]
    lower = torch.zeros((n, mx - mn + 2), dtype=torch.float32, device=mean.device)
    if n:
        m32, s32, q32 = _f32(mean, scale, Q)
        _lib.check(_lib.lib().gsac_calculate_cdf(runtime.context(mean.device), m32.data_ptr(), s32.data_ptr(), q32.data_ptr(),
                                                  n, mn, mx, lower.data_ptr(), runtime.stream_ptr(mean.device)))
    return lower


def arithmetic_encode(sym, cdf, chunk_size, N, Lp):
    _chk(sym, "sym"); _chk(cdf, "cdf")
    if sym.dim() != 1:
        raise RuntimeError(f"Expected sym to have 1 dimension, but got {sym.dim()}")
    if cdf.dim() != 2:
        raise RuntimeError(f"Expected cdf to have 2 dimensions, but got {cdf.dim()}")
    sym = sym.to(torch.int16).contiguous()
    cdf = cdf.to(torch.float32).contiguous()
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().gsac_encode(runtime.context(sym.device), sym.data_ptr(), cdf.data_ptr(), int(chunk_size), int(N), int(Lp),
                                      C.byref(pb), C.byref(nb), C.byref(pc), C.byref(nc), runtime.stream_ptr(sym.device)))
    out, cnt = _owned(pb, nb.value, np.uint8), _owned(pc, nc.value, np.int32)
    return torch.from_numpy(out).to(sym.device), torch.from_numpy(cnt).to(sym.device)


def encode_const_row(sym, row, chunk_size):
    """arithmetic_encode with the SAME CDF row for every symbol (row: 2 .. 4 floats): byte-identical to arithmetic_encode on the table that repeats the
    row -- without the table (gsac_encode_const).  Returns (bytes, cnt) as numpy arrays."""
    _chk(sym, "sym")
    sym = sym.to(torch.int16).contiguous()
    r = (C.c_float * len(row))(*[float(v) for v in row])
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().gsac_encode_const(runtime.context(sym.device), sym.data_ptr(), C.cast(r, C.c_void_p), int(chunk_size), int(sym.numel()), len(row),
                                            C.byref(pb), C.byref(nb), C.byref(pc), C.byref(nc), runtime.stream_ptr(sym.device)))
    return _owned(pb, nb.value, np.uint8), _owned(pc, nc.value, np.int32)


def _check_cnt(cnt, lengths, chunk_size, what):
    """The chunk byte-count table of a `.b` file comes off disk, and the library reads cnt[c] for every chunk c of every stream it is
    asked to decode (arithmetic_kernel.cu:290-303 does the same): a table shorter than ceil(n / chunk_size) entries per stream would be
    read past its end on the host.  Checked here, at the one layer every decoder goes through, before the pointer crosses the C ABI."""
    cs = int(chunk_size)
    if cs <= 0:
        raise ValueError(f"{what}: chunk_size must be positive")
    want = sum(-(-int(n) // cs) for n in lengths)
    if int(cnt.size) != want:
        raise ValueError(f"{what}: the chunk table has {int(cnt.size)} entries, {want} chunks of {cs} symbols are coded (corrupt or truncated .b file)")
    if cnt.size and int(cnt.min()) < 0:
        raise ValueError(f"{what}: negative chunk byte count (corrupt .b file)")


def decode_const_row(row, data, cnt, chunk_size, N, device):
    """Inverse of encode_const_row (data, cnt: numpy arrays); int16 symbols on `device`."""
    data = np.ascontiguousarray(data, dtype=np.uint8); cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    _check_cnt(cnt, [N], chunk_size, "decode_const_row")
    r = (C.c_float * len(row))(*[float(v) for v in row])
    out = torch.zeros(int(N), dtype=torch.int16, device=device)
    _lib.check(_lib.lib().gsac_decode_const(runtime.context(out.device), C.cast(r, C.c_void_p), data.ctypes.data, data.size, cnt.ctypes.data, int(chunk_size),
                                            int(N), len(row), out.data_ptr(), runtime.stream_ptr(out.device)))
    return out


def arithmetic_decode(cdf, in_cache_all, in_cnt_all, chunk_size, N, Lp):
    _chk(cdf, "cdf")
    cdf = cdf.to(torch.float32).contiguous()
    data = in_cache_all.detach().cpu().numpy().astype(np.uint8, copy=False)
    cnt = in_cnt_all.detach().cpu().numpy().astype(np.int32, copy=False)
    data = np.ascontiguousarray(data); cnt = np.ascontiguousarray(cnt)
    _check_cnt(cnt, [N], chunk_size, "arithmetic_decode")
    out = torch.zeros(int(N), dtype=torch.int16, device=cdf.device)
    _lib.check(_lib.lib().gsac_decode(runtime.context(cdf.device), cdf.data_ptr(), data.ctypes.data, data.size, cnt.ctypes.data, int(chunk_size),
                                      int(N), int(Lp), out.data_ptr(), runtime.stream_ptr(cdf.device)))
    return out


def encode_gaussian(x, mean, scale, Q, chunk_size):
    """round(x / Q) -> symbols -> chunked range coder with the Gaussian CDF entries evaluated on the fly.
    Byte-identical to calculate_cdf + arithmetic_encode (encodings_cuda.py:336-371)."""
    for t, nm in ((x, "x"), (mean, "mean"), (scale, "scale"), (Q, "Q")):
        _chk(t, nm)
    n = int(x.shape[0])
    mn, mx = C.c_float(), C.c_float()
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    x32, m32, s32, q32 = _f32(x, mean, scale, Q)
    _lib.check(_lib.lib().gsac_encode_gaussian(runtime.context(x.device), x32.data_ptr(), m32.data_ptr(), s32.data_ptr(),
                                               q32.data_ptr(), n, int(chunk_size), C.byref(mn), C.byref(mx), C.byref(pb), C.byref(nb),
                                               C.byref(pc), C.byref(nc), runtime.stream_ptr(x.device)))
    out, cnt = _owned(pb, nb.value, np.uint8), _owned(pc, nc.value, np.int32)
    return mn.value, mx.value, torch.from_numpy(out).to(x.device), torch.from_numpy(cnt).to(x.device)


def decode_gaussian(mean, scale, Q, min_value, max_value, in_cache_all, in_cnt_all, chunk_size):
    """Inverse of encode_gaussian: (sym + min) * Q, float32 on mean.device (encodings_cuda.py:399-433)."""
    for t, nm in ((mean, "mean"), (scale, "scale"), (Q, "Q")):
        _chk(t, nm)
    data = np.ascontiguousarray(in_cache_all.detach().cpu().numpy().astype(np.uint8, copy=False))
    cnt = np.ascontiguousarray(in_cnt_all.detach().cpu().numpy().astype(np.int32, copy=False))
    n = int(mean.shape[0])
    _check_cnt(cnt, [n], chunk_size, "decode_gaussian")
    out = torch.empty(n, dtype=torch.float32, device=mean.device)
    m32, s32, q32 = _f32(mean, scale, Q)
    _lib.check(_lib.lib().gsac_decode_gaussian(runtime.context(mean.device), m32.data_ptr(), s32.data_ptr(), q32.data_ptr(), n,
                                               float(min_value), float(max_value), data.ctypes.data, data.size, cnt.ctypes.data, int(chunk_size),
                                               out.data_ptr(), runtime.stream_ptr(mean.device)))
    return out


def encode_gaussian_slices(x, mean, scale, Q, slice_start, chunk_size):
    """All slices [slice_start[s], slice_start[s+1]) of an attribute in one call (gsac_encode_gaussian_slices).
    Returns (mins, maxs, bytes, cnt) as numpy arrays: float32 (nslices) x 2, uint8, int32 (chunks, slice by slice)."""
    for t, nm in ((x, "x"), (mean, "mean"), (scale, "scale"), (Q, "Q")):
        _chk(t, nm)
    ss = np.ascontiguousarray(np.asarray(slice_start, dtype=np.int64))
    ns = ss.size - 1
    mins, maxs = np.empty(ns, dtype=np.float32), np.empty(ns, dtype=np.float32)
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    x32, m32, s32, q32 = _f32(x, mean, scale, Q)
    _lib.check(_lib.lib().gsac_encode_gaussian_slices(runtime.context(x.device), x32.data_ptr(), m32.data_ptr(), s32.data_ptr(),
                                                      q32.data_ptr(), ss.ctypes.data, ns, int(chunk_size), mins.ctypes.data, maxs.ctypes.data,
                                                      C.byref(pb), C.byref(nb), C.byref(pc), C.byref(nc), runtime.stream_ptr(x.device)))
    out, cnt = _owned(pb, nb.value, np.uint8), _owned(pc, nc.value, np.int32)
    return mins, maxs, out, cnt


def decode_gaussian_slices(mean, scale, Q, slice_start, mins, maxs, data, cnt, chunk_size):
    """Inverse of encode_gaussian_slices; data / cnt are the concatenated payloads / chunk byte counts (numpy)."""
    for t, nm in ((mean, "mean"), (scale, "scale"), (Q, "Q")):
        _chk(t, nm)
    ss = np.ascontiguousarray(np.asarray(slice_start, dtype=np.int64))
    mins = np.ascontiguousarray(mins, dtype=np.float32); maxs = np.ascontiguousarray(maxs, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.uint8); cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    _check_cnt(cnt, np.diff(ss), chunk_size, "decode_gaussian_slices")
    if mins.size != ss.size - 1 or maxs.size != ss.size - 1:
        raise ValueError("decode_gaussian_slices: one (min, max) per slice")
    out = torch.empty(int(ss[-1]), dtype=torch.float32, device=mean.device)
    m32, s32, q32 = _f32(mean, scale, Q)
    _lib.check(_lib.lib().gsac_decode_gaussian_slices(runtime.context(mean.device), m32.data_ptr(), s32.data_ptr(), q32.data_ptr(),
                                                      ss.ctypes.data, ss.size - 1, mins.ctypes.data, maxs.ctypes.data, data.ctypes.data, data.size,
                                                      cnt.ctypes.data, int(chunk_size), out.data_ptr(), runtime.stream_ptr(mean.device)))
    return out


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _mix_args(mean_list, scale_list, prob_list, Q):
    k = len(mean_list)
    if not (1 <= k <= 4) or len(scale_list) != k or len(prob_list) != k:
        raise RuntimeError("a mixture has 1..4 components with one mean, scale and prob tensor each")
    for i, (m, s, p) in enumerate(zip(mean_list, scale_list, prob_list)):
        for t, nm in ((m, f"mean[{i}]"), (s, f"scale[{i}]"), (p, f"prob[{i}]")):
            _chk(t, nm)
    _chk(Q, "Q")
    keep = _f32(*mean_list, *scale_list, *prob_list, Q)      # alive until the library call has returned
    return k, keep, _ptr_array(keep[:k]), _ptr_array(keep[k:2 * k]), _ptr_array(keep[2 * k:3 * k]), keep[3 * k]


def calculate_cdf_mixed(mean_list, scale_list, prob_list, Q, min_value, max_value):
    """The mixture's (n, max-min+2) table: sum of calculate_cdf(mean_c, scale_c, Q) * prob_c over the components, clamped."""
    k, keep, pm, ps, pp, q32 = _mix_args(mean_list, scale_list, prob_list, Q)
    mn, mx = int(min_value), int(max_value)
    n = int(q32.shape[0])
    lower = torch.zeros((n, mx - mn + 2), dtype=torch.float32, device=q32.device)
    if n:
        _lib.check(_lib.lib().gsac_calculate_cdf_mixed(runtime.context(q32.device), pm, ps, pp, k, q32.data_ptr(), n, mn, mx, lower.data_ptr(),
                                                        runtime.stream_ptr(q32.device)))
    return lower


def encode_gaussian_mixed(x, mean_list, scale_list, prob_list, Q, chunk_size):
    """round(x / Q) -> symbols -> chunked range coder, the mixture's CDF entries evaluated on the fly.
    Byte-identical to calculate_cdf_mixed + arithmetic_encode."""
    _chk(x, "x")
    k, keep, pm, ps, pp, q32 = _mix_args(mean_list, scale_list, prob_list, Q)
    (x32,) = _f32(x)
    n = int(x32.shape[0])
    mn, mx = C.c_float(), C.c_float()
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().gsac_encode_gaussian_mixed(runtime.context(x.device), x32.data_ptr(), pm, ps, pp, k, q32.data_ptr(), n, int(chunk_size),
                                                     C.byref(mn), C.byref(mx), C.byref(pb), C.byref(nb), C.byref(pc), C.byref(nc),
                                                     runtime.stream_ptr(x.device)))
    out, cnt = _owned(pb, nb.value, np.uint8), _owned(pc, nc.value, np.int32)
    return mn.value, mx.value, torch.from_numpy(out).to(x.device), torch.from_numpy(cnt).to(x.device)


def decode_gaussian_mixed(mean_list, scale_list, prob_list, Q, min_value, max_value, in_cache_all, in_cnt_all, chunk_size):
    """Inverse of encode_gaussian_mixed: (sym + min) * Q, float32 on the parameters' device."""
    k, keep, pm, ps, pp, q32 = _mix_args(mean_list, scale_list, prob_list, Q)
    data = np.ascontiguousarray(in_cache_all.detach().cpu().numpy().astype(np.uint8, copy=False))
    cnt = np.ascontiguousarray(in_cnt_all.detach().cpu().numpy().astype(np.int32, copy=False))
    n = int(q32.shape[0])
    _check_cnt(cnt, [n], chunk_size, "decode_gaussian_mixed")
    out = torch.empty(n, dtype=torch.float32, device=q32.device)
    _lib.check(_lib.lib().gsac_decode_gaussian_mixed(runtime.context(q32.device), pm, ps, pp, k, q32.data_ptr(), n, float(min_value), float(max_value),
                                                     data.ctypes.data, data.size, cnt.ctypes.data, int(chunk_size), out.data_ptr(),
                                                     runtime.stream_ptr(q32.device)))
    return out


def encode_gaussian_mixed_slices(x, mean_list, scale_list, prob_list, Q, slice_start, chunk_size):
    """All slices of one mixture-coded attribute in one call (gsac_encode_gaussian_mixed_slices): HAC++ codes `feat` as five
    ten-channel groups per 3000-anchor slice, each group under a two-component mixture (HAC-plus/scene/gaussian_model.py:1300-1321).
    Returns (mins, maxs, bytes, cnt) as numpy arrays, as encode_gaussian_slices does."""
    _chk(x, "x")
    k, keep, pm, ps, pp, q32 = _mix_args(mean_list, scale_list, prob_list, Q)
    (x32,) = _f32(x)
    ss = np.ascontiguousarray(np.asarray(slice_start, dtype=np.int64))
    ns = ss.size - 1
    mins, maxs = np.empty(ns, dtype=np.float32), np.empty(ns, dtype=np.float32)
    pb, nb, pc, nc = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().gsac_encode_gaussian_mixed_slices(runtime.context(x.device), x32.data_ptr(), pm, ps, pp, k, q32.data_ptr(), ss.ctypes.data, ns,
                                                            int(chunk_size), mins.ctypes.data, maxs.ctypes.data, C.byref(pb), C.byref(nb), C.byref(pc),
                                                            C.byref(nc), runtime.stream_ptr(x.device)))
    out, cnt = _owned(pb, nb.value, np.uint8), _owned(pc, nc.value, np.int32)
    return mins, maxs, out, cnt


def decode_gaussian_mixed_slices(mean_list, scale_list, prob_list, Q, slice_start, mins, maxs, data, cnt, chunk_size):
    """Inverse of encode_gaussian_mixed_slices."""
    k, keep, pm, ps, pp, q32 = _mix_args(mean_list, scale_list, prob_list, Q)
    ss = np.ascontiguousarray(np.asarray(slice_start, dtype=np.int64))
    mins = np.ascontiguousarray(mins, dtype=np.float32); maxs = np.ascontiguousarray(maxs, dtype=np.float32)
    data = np.ascontiguousarray(data, dtype=np.uint8); cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    _check_cnt(cnt, np.diff(ss), chunk_size, "decode_gaussian_mixed_slices")
    if mins.size != ss.size - 1 or maxs.size != ss.size - 1:
        raise ValueError("decode_gaussian_mixed_slices: one (min, max) per slice")
    out = torch.empty(int(ss[-1]), dtype=torch.float32, device=q32.device)
    _lib.check(_lib.lib().gsac_decode_gaussian_mixed_slices(runtime.context(q32.device), pm, ps, pp, k, q32.data_ptr(), ss.ctypes.data, ss.size - 1,
                                                            mins.ctypes.data, maxs.ctypes.data, data.ctypes.data, data.size, cnt.ctypes.data,
                                                            int(chunk_size), out.data_ptr(), runtime.stream_ptr(q32.device)))
    return out
