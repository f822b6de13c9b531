"""Drop-in for the reference's plugin API (identical file in HAC / HAC-plus / TC-GS /
CAT-3DGS: src/gs_compress/*/utils/pcc_utils.py).

Same names, arguments, defaults and returned dict keys as the reference
(pcc_utils.py:12-22, 24-217, 230-400); the work happens in libgauspcc.so on the
MI355X.  There is no CPU path: without the HIP library or a GPU these functions raise.

Deliberate differences, all on the error side:
  * duplicate input points raise (the reference silently corrupts occupancy);
  * a bare file name as output_path works (reference: os.makedirs('') raises, :47);
  * decompress_point_cloud(output_path=...) really writes the ASCII PLY (reference: NameError,
    `io` is never imported, :392).
The container written by default is the chunked one (version 4: per-level chunk sizes, two coder lanes per byte-counted
chunk, a carry-propagating range coder in the lanes, parallel decode; HISTORY.md section 5); chunk_log2=0 writes the
reference's exact layout.  Every layout this library ever wrote (versions 1-4) and the reference's are read back transparently;
`gauspcc_amd.pcc_utils.CONTAINER_VERSION = 3` (or GAUSPCC_CONTAINER_VERSION=3) keeps writing round 3's layout.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib, runtime

DEFAULT_CHUNK_LOG2 = int(os.environ.get("GAUSPCC_CHUNK_LOG2", "11"))
CONTAINER_VERSION = int(os.environ.get("GAUSPCC_CONTAINER_VERSION", "4"))   # chunked containers: 4, or 3 (torchac's coder in the lanes: round 3's files)


def _set_version(ctx, version=None):
    _lib.check(_lib.lib().gpcc_ctx_set_container_version(ctx, CONTAINER_VERSION if version is None else int(version)))

_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3}


def calculate_morton_order(x: torch.Tensor) -> torch.Tensor:
    """Calculate Morton order of the input points.

    (Reference semantics, pcc_utils.py:12-22: despite the name this is the raster order
    of x + y*M + z*M^2 after the per-axis min shift.)  Returns a LongTensor permutation on
    x.device.  The reference round-trips through the host and np.argsort; here key build
    and the radix sort run on the device.
    """
    assert len(x.shape) == 2 and x.shape[1] == 3, f'Input data must be a 3D point cloud, but got {x.shape}.'
    if not x.is_cuda:
        raise RuntimeError("gauspcc_amd.calculate_morton_order needs a tensor on the MI355X (no CPU path)")
    xs = x.detach()
    if xs.dtype not in _DTYPES:
        xs = xs.to(torch.float32) if xs.is_floating_point() else xs.to(torch.int64)
    xs = xs.contiguous()
    out = torch.empty(xs.shape[0], dtype=torch.int64, device=xs.device)
    if xs.shape[0] == 0:
        return out
    ctx = runtime.context(xs.device)
    _lib.check(_lib.lib().gpcc_raster_order(ctx, xs.data_ptr(), _DTYPES[xs.dtype], xs.shape[0], out.data_ptr(), runtime.stream_ptr(xs.device)))
    return out


def voxelise(xyz: torch.Tensor, is_data_pre_quantized: bool = True, posQ=1) -> torch.Tensor:
    """The quantisation in front of the codec, on the device and in the tensor's own dtype
    (compress_ue_4stage_conv.py:89-94): `xyz / 0.001 + 131072` unless the data is pre-quantised, then
    `torch.round(xyz / posQ).int()`.  xyz: (N,3) float32 / float64 tensor on the MI355X -> (N,3) int32 on the same device.
    (HAC's `torch.round(anchor / voxel_size)`, gaussian_model.py:1107, is voxelise(anchor / voxel_size).)"""
    assert len(xyz.shape) == 2 and xyz.shape[1] == 3
    if not xyz.is_cuda:
        raise RuntimeError("gauspcc_amd.voxelise needs a tensor on the MI355X (no CPU path)")
    x = xyz.detach()
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float64 if x.dtype == torch.int64 else torch.float32)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    flags = (0 if is_data_pre_quantized else 1) | 2
    _lib.check(_lib.lib().gpcc_voxelise(runtime.context(x.device), x.data_ptr(), _DTYPES[x.dtype], x.shape[0], 0.001, 131072.0, float(posQ), flags,
                                        out.data_ptr(), runtime.stream_ptr(x.device)))
    return out


def _encode_to_bytes(xyz_int32: torch.Tensor, model, chunk_log2: int, posQ, ideal_bits: bool = False, version=None):
    ctx = runtime.context(xyz_int32.device)
    _set_version(ctx, version)
    pb, nb, st = C.c_void_p(), C.c_int64(), _lib.Stats()
    st.flags = 1 if ideal_bits else 0   # GPCC_STATS_IDEAL_BITS: the bpp estimator next to the coder (diagnostic, ~4 % of an encode)
    _lib.check(_lib.lib().gpcc_encode(ctx, model.handle, xyz_int32.data_ptr(), xyz_int32.shape[0], chunk_log2, runtime.f16_bits(posQ),
                                      C.byref(pb), C.byref(nb), C.byref(st), runtime.stream_ptr(xyz_int32.device)))
    data = C.string_at(pb, nb.value)
    return data, st


def compress_point_cloud(
    xyz_quantized,            # Quantized point cloud coordinates, numpy array or torch tensor
    ckpt_path,                # Path to pre-trained weights file
    output_path,              # Output bin file path
    channels=32,              # Network channel count
    kernel_size=5,            # Convolution kernel size
    posQ=1,                   # Quantization scale
    *,
    chunk_log2=None,          # extension: 0 = reference container, 6..14 = largest chunk size of the chunked container (default 11)
):
    """Compress point cloud into a bin file (reference: pcc_utils.py:24-217).

    Returns {'bpp', 'enc_time', 'file_size_bits', 'num_points', 'output_path'}; enc_time
    covers the same span as the reference (:78-189: octree build .. packed byte stream,
    excluding model load and the file write).
    """
    d = os.path.dirname(output_path)
    if d:
        os.makedirs(d, exist_ok=True)
    if chunk_log2 is None:
        chunk_log2 = DEFAULT_CHUNK_LOG2
    if isinstance(xyz_quantized, np.ndarray):
        xyz = torch.tensor(xyz_quantized)
    else:
        xyz = xyz_quantized.clone()
    if not torch.cuda.is_available():
        raise RuntimeError("gauspcc_amd.compress_point_cloud needs an MI355X (no CPU path)")
    device = xyz.device if xyz.is_cuda else torch.device('cuda', torch.cuda.current_device())
    model = runtime.get_model(ckpt_path, channels, kernel_size, device)
    N = xyz.shape[0]
    xyz = xyz.to(device).int().contiguous()          # reference: torch.cat(...).int()  (:73)

    # the reference brackets its span with device-wide syncs (:78, :189); the caller's stream is what this call's work is
    # ordered on, and syncing only that lets other host threads keep their own calls in flight on the same GPU
    torch.cuda.current_stream(device).synchronize()
    enc_time_start = time.time()
    data, st = _encode_view(xyz, model, chunk_log2, posQ)   # the context's pinned output buffer, written to the file as it is
    torch.cuda.current_stream(device).synchronize()
    enc_time_end = time.time()

    with open(output_path, 'wb') as f:
        f.write(data)

    enc_time = enc_time_end - enc_time_start
    file_size_bits = os.stat(output_path).st_size * 8
    bpp = file_size_bits / N
    return {
        'bpp': bpp,
        'enc_time': enc_time,
        'file_size_bits': file_size_bits,
        'num_points': N,
        'output_path': output_path,
    }


def save_ply_ascii_geo(coords, filedir):
    """ASCII PLY writer with the layout of kit/io.py:36-49 (without the open3d dependency)."""
    coords = np.asarray(coords, dtype=np.float32)
    with open(filedir, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write("element vertex " + str(coords.shape[0]) + "\n")
        f.write("property float x\nproperty float y\nproperty float z\n")
        f.write("end_header\n")
        for p in coords:
            f.write(f"{p[0]} {p[1]} {p[2]}\n")


def _encode_view(xyz_int32: torch.Tensor, model, chunk_log2: int, posQ):
    """gpcc_encode without the copy into a Python bytes object: a ctypes byte array over the context's pinned output buffer.
    The view is valid only until the NEXT call of any kind that writes that buffer on this context: gpcc_encode, gpcc_rc_encode
    and every gsac_encode* (the attribute coders of arithmetic.py / encodings_cuda.py share it).  Write it to a file or hand
    it to _decode_bytes first; _encode_to_bytes returns an owned copy."""
    ctx = runtime.context(xyz_int32.device)
    _set_version(ctx)
    pb, nb, st = C.c_void_p(), C.c_int64(), _lib.Stats()
    _lib.check(_lib.lib().gpcc_encode(ctx, model.handle, xyz_int32.data_ptr(), xyz_int32.shape[0], chunk_log2, runtime.f16_bits(posQ),
                                      C.byref(pb), C.byref(nb), C.byref(st), runtime.stream_ptr(xyz_int32.device)))
    return (C.c_ubyte * nb.value).from_address(pb.value), st


def _header_points(head: bytes, nbytes: int = None):
    """Point count from a chunked container's header (FF FF | version | chunk_log2 | posQ | L | 0 | u32 n[L] | u32 N), else None.
    The header is untrusted and this count sizes a device allocation BEFORE the library's own checks run, so the same cheap bounds
    apply here (csrc/container.hpp: container_precheck): the last level must be able to expand to the count (1..8 children per node),
    every level at most 8x the one above, and no level may claim more nodes than the file could code (a node costs at least ~2^-13
    bytes: all nodes <= nbytes << 13).  Anything else is reported as None, and the caller takes the path on which the library sizes the
    output from what it decoded."""
    if len(head) >= 12 and head[0] == 0xFF and head[1] == 0xFF and 1 <= head[6] <= 21 and len(head) >= 12 + 4 * head[6]:
        L = head[6]
        npts = int.from_bytes(head[8 + 4 * L: 12 + 4 * L], "little")
        lv = [int.from_bytes(head[8 + 4 * d: 12 + 4 * d], "little") for d in range(L)]
        if lv[0] <= 0 or any(b <= 0 or b > 8 * a for a, b in zip(lv, lv[1:])):
            return None
        if nbytes is not None and sum(lv) > (int(nbytes) << 13):
            return None
        if lv[-1] <= npts <= 8 * lv[-1]:
            return npts
    return None


def _decode_bytes(data, model, device):
    """data: bytes, or a ctypes byte array (e.g. from _encode_view).  The chunked containers announce their point count,
    so the output tensor is allocated up front and the library writes the points straight into it."""
    ctx = runtime.context(device)
    n, pq, st = C.c_int64(), C.c_uint16(), _lib.Stats()
    if isinstance(data, C.Array):
        ptr, nbytes, head = C.c_void_p(C.addressof(data)), len(data), bytes(data[:96])
    else:
        if not isinstance(data, bytes):
            data = bytes(data)
        ptr, nbytes, head = C.cast(C.c_char_p(data), C.c_void_p), len(data), data[:96]   # the bytes object's own storage: only read
    npts = _header_points(head, nbytes)
    if npts is not None and 0 < npts < (1 << 31):
        out = torch.empty((npts, 3), dtype=torch.int32, device=device)
        _lib.check(_lib.lib().gpcc_decode_to(ctx, model.handle, ptr, nbytes, out.data_ptr(), npts, C.byref(n), C.byref(pq),
                                             C.byref(st), runtime.stream_ptr(device)))
        return out[: n.value], runtime.bits_f16(pq.value), st
    px = C.c_void_p()
    _lib.check(_lib.lib().gpcc_decode(ctx, model.handle, ptr, nbytes, C.byref(px), C.byref(n), C.byref(pq),
                                      C.byref(st), runtime.stream_ptr(device)))
    out = torch.empty((n.value, 3), dtype=torch.int32, device=device)
    if n.value:
        # context-owned device buffer -> caller-owned tensor (D2D on the current stream)
        _lib.check(_lib.lib().gpcc_memcpy_d2d(ctx, out.data_ptr(), px, 12 * n.value, runtime.stream_ptr(device)))
    return out, runtime.bits_f16(pq.value), st


def decompress_point_cloud(
    bin_file_path,           # Path to compressed bin file
    ckpt_path,               # Path to pre-trained weights file
    output_path=None,        # Path for output ply file (optional)
    channels=32,             # Network channel count
    kernel_size=5,           # Convolution kernel size
    is_data_pre_quantized=True  # Whether original point cloud is pre-quantized
):
    """Decompress point cloud from bin file (reference: pcc_utils.py:230-400).

    Returns {'dec_time', 'num_points', 'point_cloud', 'output_path'}; point_cloud is an
    (N,3) float tensor on the device in the reference's decoder order (callers re-sort
    with calculate_morton_order).
    """
    if output_path:
        d = os.path.dirname(output_path)
        if d:
            os.makedirs(d, exist_ok=True)
    if not torch.cuda.is_available():
        raise RuntimeError("gauspcc_amd.decompress_point_cloud needs an MI355X (no CPU path)")
    device = torch.device('cuda', torch.cuda.current_device())
    model = runtime.get_model(ckpt_path, channels, kernel_size, device)
    with open(bin_file_path, 'rb') as f:
        data = f.read()

    torch.cuda.current_stream(device).synchronize()
    dec_time_start = time.time()
    with torch.no_grad():
        scan, posQ, st = _decode_bytes(data, model, device)
        if is_data_pre_quantized:
            scan = scan * posQ.item()                       # :378-379
        else:
            scan = (scan * posQ.item() - 131072) * 0.001    # :381
    torch.cuda.current_stream(device).synchronize()
    dec_time_end = time.time()
    dec_time = dec_time_end - dec_time_start

    point_cloud = scan
    if output_path:
        save_ply_ascii_geo(point_cloud.cpu().numpy(), output_path)
    return {
        'dec_time': dec_time,
        'num_points': point_cloud.shape[0],
        'point_cloud': point_cloud,
        'output_path': output_path,
    }


# ---------------------------------------------------------------------------------------------------------------- batches
# The reference's coordinate tensor has a batch column (coords = [b, x, y, z], pcc_utils.py:73, b pinned to 0; sort_CF orders by
# batch last, kit/op.py:17-30) and its stand-alone CLI loops over files (compress_ue_4stage_conv.py:72-75).  A scene is a chain of
# ~60 dependent launches per octree level whatever its size, so scenes of 10^4 .. 10^5 anchors leave most of an MI355X idle;
# the functions below code K scenes through ONE chain of launches (csrc/forest.hpp).  Every scene's file is byte-identical to
# what compress_point_cloud writes for it alone, and any reader of that file reads these.

def _encode_batch(xyz_list, model, chunk_log2: int, posQ_list, version=None, view=False):
    """xyz_list: (N_i, 3) int32 tensors on one device -> ([bytes, ...], [Stats, ...], batched flag).  view=True: ctypes byte
    arrays over the context's pinned output buffer instead of owned bytes (valid until the context's next encode of any kind)."""
    K = len(xyz_list)
    device = xyz_list[0].device
    ctx = runtime.context(device)
    _set_version(ctx, version)
    ptrs = (C.c_void_p * K)(*[x.data_ptr() for x in xyz_list])
    ns = (C.c_int64 * K)(*[int(x.shape[0]) for x in xyz_list])
    pq = (C.c_uint16 * K)(*[runtime.f16_bits(q) for q in posQ_list])
    offs = (C.c_int64 * (K + 1))()
    stats = (_lib.Stats * K)()
    pb, batched = C.c_void_p(), C.c_int(0)
    _lib.check(_lib.lib().gpcc_encode_batch(ctx, model.handle, ptrs, ns, K, chunk_log2, pq, C.byref(pb), offs, stats, C.byref(batched), runtime.stream_ptr(device)))
    if view:
        return [(C.c_ubyte * (offs[i + 1] - offs[i])).from_address(pb.value + offs[i]) for i in range(K)], list(stats), bool(batched.value)
    blob = C.string_at(pb, offs[K])
    return [blob[offs[i]:offs[i + 1]] for i in range(K)], list(stats), bool(batched.value)


def _decode_batch(datas, model, device):
    """datas: container bytes of K scenes -> ([(N_i, 3) int32 tensors], [posQ], [Stats], batched flag)."""
    K = len(datas)
    datas = [d if isinstance(d, (bytes, C.Array)) else bytes(d) for d in datas]
    ctx = runtime.context(device)
    npts = [_header_points(bytes(d[:96]), len(d)) for d in datas]
    if any(n is None or not (0 < n < (1 << 31)) for n in npts):
        # a container without a point count (the reference layout): one by one
        outs, pqs, sts = [], [], []
        for d in datas:
            o, q, s = _decode_bytes(d, model, device)
            outs.append(o); pqs.append(q); sts.append(s)
        return outs, pqs, sts, False
    outs = [torch.empty((n, 3), dtype=torch.int32, device=device) for n in npts]
    bptr = (C.c_void_p * K)(*[C.c_void_p(C.addressof(d)) if isinstance(d, C.Array) else C.cast(C.c_char_p(d), C.c_void_p) for d in datas])
    bn = (C.c_int64 * K)(*[len(d) for d in datas])
    optr = (C.c_void_p * K)(*[o.data_ptr() for o in outs])
    caps = (C.c_int64 * K)(*npts)
    n_out = (C.c_int64 * K)()
    pq = (C.c_uint16 * K)()
    stats = (_lib.Stats * K)()
    batched = C.c_int(0)
    _lib.check(_lib.lib().gpcc_decode_batch(ctx, model.handle, bptr, bn, K, optr, caps, n_out, pq, stats, C.byref(batched), runtime.stream_ptr(device)))
    return [o[: n_out[i]] for i, o in enumerate(outs)], [runtime.bits_f16(pq[i]) for i in range(K)], list(stats), bool(batched.value)


def compress_point_clouds(
    xyz_quantized_list,       # list of quantized point clouds (numpy arrays or torch tensors)
    ckpt_path,                # Path to pre-trained weights file
    output_paths,             # one bin file path per cloud
    channels=32,
    kernel_size=5,
    posQ=1,                   # one scale for all clouds, or a list
    *,
    chunk_log2=None,
):
    """compress_point_cloud (pcc_utils.py:24-217) over a batch: the clouds share every launch of an octree depth.  Returns one
    dict per cloud with the reference's keys; 'enc_time' is the batch's span divided by the number of clouds, 'batch_enc_time'
    the span itself."""
    assert len(xyz_quantized_list) == len(output_paths) and len(output_paths) > 0
    if chunk_log2 is None:
        chunk_log2 = DEFAULT_CHUNK_LOG2
    if not torch.cuda.is_available():
        raise RuntimeError("gauspcc_amd.compress_point_clouds needs an MI355X (no CPU path)")
    first = xyz_quantized_list[0]
    device = first.device if (isinstance(first, torch.Tensor) and first.is_cuda) else torch.device('cuda', torch.cuda.current_device())
    model = runtime.get_model(ckpt_path, channels, kernel_size, device)
    xs = [(torch.tensor(x) if isinstance(x, np.ndarray) else x).to(device).int().contiguous() for x in xyz_quantized_list]
    posQs = list(posQ) if isinstance(posQ, (list, tuple)) else [posQ] * len(xs)
    for p in output_paths:
        d = os.path.dirname(p)
        if d:
            os.makedirs(d, exist_ok=True)
    torch.cuda.current_stream(device).synchronize()
    t0 = time.time()
    blobs, _, _ = _encode_batch(xs, model, chunk_log2, posQs)
    torch.cuda.current_stream(device).synchronize()
    span = time.time() - t0
    out = []
    for x, p, b in zip(xs, output_paths, blobs):
        with open(p, 'wb') as f:
            f.write(b)
        bits = os.stat(p).st_size * 8
        out.append({'bpp': bits / x.shape[0], 'enc_time': span / len(xs), 'batch_enc_time': span, 'file_size_bits': bits, 'num_points': x.shape[0], 'output_path': p})
    return out


def decompress_point_clouds(bin_file_paths, ckpt_path, output_paths=None, channels=32, kernel_size=5, is_data_pre_quantized=True):
    """decompress_point_cloud (pcc_utils.py:230-400) over a batch of files; one dict per file with the reference's keys."""
    if not torch.cuda.is_available():
        raise RuntimeError("gauspcc_amd.decompress_point_clouds needs an MI355X (no CPU path)")
    device = torch.device('cuda', torch.cuda.current_device())
    model = runtime.get_model(ckpt_path, channels, kernel_size, device)
    datas = []
    for p in bin_file_paths:
        with open(p, 'rb') as f:
            datas.append(f.read())
    torch.cuda.current_stream(device).synchronize()
    t0 = time.time()
    with torch.no_grad():
        scans, posQs, _, _ = _decode_batch(datas, model, device)
        clouds = [s * q.item() if is_data_pre_quantized else (s * q.item() - 131072) * 0.001 for s, q in zip(scans, posQs)]   # :378-381
    torch.cuda.current_stream(device).synchronize()
    span = time.time() - t0
    out = []
    for i, pc in enumerate(clouds):
        op = output_paths[i] if output_paths else None
        if op:
            d = os.path.dirname(op)
            if d:
                os.makedirs(d, exist_ok=True)
            save_ply_ascii_geo(pc.cpu().numpy(), op)
        out.append({'dec_time': span / len(clouds), 'batch_dec_time': span, 'num_points': pc.shape[0], 'point_cloud': pc, 'output_path': op})
    return out
