"""The HAC attribute codec wrappers with the reference's names, arguments and `.b` file
formats (src/gs_compress/HAC/utils/encodings_cuda.py:317-492), on top of
gauspcc_amd.arithmetic instead of the CUDA `arithmetic` extension.

    encoder_gaussian(_chunk) / decoder_gaussian(_chunk)   :317-433
    encoder / decoder (Bernoulli, one global p)            :435-492
    encoder_factorized(_chunk) / decoder_factorized(_chunk) :38-175   (a learned per-channel density, `lower_func`)
    encoder_gaussian_mixed(_chunk) / decoder_gaussian_mixed(_chunk)   HAC-plus/utils/encodings_cuda.py:177-317 (HAC++'s
                                                           K-component mixture; callers HAC-plus/scene/gaussian_model.py:1315, 1499)

encoder_gaussian / decoder_gaussian go through the fused gsac_encode_gaussian / gsac_decode_gaussian: the reference
writes and re-reads an (n, max-min+2) float table per slice (36 MB for a 3000-anchor feature slice, 334 slices per
million anchors); here the CDF entries are evaluated inside the coder.  Same `.b` bytes either way
(tests/test_gpu_attributes.py::test_fused_gaussian_matches_table_path).
"""
import os

import numpy as np
import torch

from . import arithmetic

chunk_size_cuda = 10000


def _write_one(job):
    with open(job[0], 'wb') as fout:
        fout.write(job[1])


import threading as _threading

_tls = _threading.local()      # .deferred: the calling thread's active deferred_writes context, if any (per thread: contexts of two threads must not see each other)


def _write_files_now(jobs):
    import ctypes as C
    from . import _lib
    n = len(jobs)
    paths = (C.c_char_p * n)(*[os.fsencode(j[0]) for j in jobs])
    blobs = [j[1] for j in jobs]                                     # kept alive for the call
    data = (C.c_char_p * n)(*blobs)
    sizes = (C.c_int64 * n)(*[len(b) for b in blobs])
    _lib.check(_lib.lib().gpcc_write_files(paths, data, sizes, n, 8))


def _write_files(jobs):
    """The per-slice `.b` files of an attribute (334 per million anchors and attribute; HAC++: 2 338 files per million anchors): written by
    libgauspcc on native threads (gpcc_write_files) -- in Python every file costs ~55 us of interpreter time under the GIL, thread pool or not
    (0.13 s of HAC++'s 0.27 s per million anchors).  Inside a `deferred_writes()` block the call is handed to a background thread (the library call
    releases the GIL) and the next attribute is coded while the files of this one go to disk."""
    if not jobs:
        return
    d = getattr(_tls, 'deferred', None)
    if d is not None:
        d.submit(jobs)
    else:
        _write_files_now(jobs)


def _read_slice_files(paths, lens=None):
    """lens: the symbol count of every slice -- a file whose chunk table is not exactly ceil(len / chunk_size_cuda) int32 entries is rejected here
    (the library reads one count per chunk: a shorter table would be read past its end, a longer one would shift every later slice's counts).
    The `[min, max, len(cnt) * 4, cnt..., payload]` files of many slices (encoder_gaussian*_slices) read by libgauspcc on native threads
    (gpcc_read_files: 2 343 files cost ~27 ms of open / read / frombuffer in Python): (mins, maxs, cnts, datas) with numpy views into one blob that
    is valid until this thread's next call -- callers concatenate them at once."""
    import ctypes as C
    from . import _lib
    n = len(paths)
    cp = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    offs = (C.c_int64 * (n + 1))()
    pb = C.c_void_p()
    _lib.check(_lib.lib().gpcc_read_files(cp, n, 8, C.byref(pb), offs))
    total = offs[n]
    blob = np.frombuffer((C.c_ubyte * max(int(total), 1)).from_address(pb.value), dtype=np.uint8)
    mins, maxs, cnts, datas = [], [], [], []
    for i in range(n):
        a, b = int(offs[i]), int(offs[i + 1])
        if b - a < 12:
            raise RuntimeError(f"{paths[i]}: truncated slice file")
        head = blob[a:a + 12]
        mins.append(head[0:4].view(np.float32)[0]); maxs.append(head[4:8].view(np.float32)[0])
        lc = int(head[8:12].view(np.int32)[0])
        if lc < 0 or lc % 4 or a + 12 + lc > b:
            raise RuntimeError(f"{paths[i]}: bad chunk table")
        if lens is not None and lc != 4 * (-(-int(lens[i]) // chunk_size_cuda)):
            raise RuntimeError(f"{paths[i]}: chunk table of {lc // 4} entries, the slice's {int(lens[i])} symbols are {-(-int(lens[i]) // chunk_size_cuda)} chunks")
        cnts.append(blob[a + 12:a + 12 + lc].view(np.int32))
        datas.append(blob[a + 12 + lc:b])
    return mins, maxs, cnts, datas


class deferred_writes:
    """`with deferred_writes():` -- the slice files written inside the block are handed to background threads and are all on disk when the block ends
    (an error of any of them is raised there).  On a real file system the 1 002 (HAC) / 2 343 (HAC++) files of a million anchors take 0.1-0.2 s, as long
    as the coding itself; the reference writes each slice's file inside its loop."""

    def __enter__(self):
        import queue
        import threading
        self._q, self._errors, self._outer = queue.Queue(), [], getattr(_tls, 'deferred', None)
        # ONE worker: attributes' files go out one attribute after the other (eight native threads each, as before) -- several attributes at once
        # contend for the directory (measured on the GPU box's overlay file system: 2 343 files 0.25 -> 0.9 s)
        self._worker = threading.Thread(target=self._run, daemon=True)
        self._worker.start()
        _tls.deferred = self
        return self

    def _run(self):
        while True:
            jobs = self._q.get()
            if jobs is None:
                return
            try:
                _write_files_now(jobs)
            except BaseException as e:      # noqa: BLE001 -- re-raised by __exit__ on the caller's thread
                self._errors.append(e)

    def submit(self, jobs):
        self._q.put(jobs)

    def __exit__(self, et, ev, tb):
        _tls.deferred = self._outer
        self._q.put(None)
        self._worker.join()
        if self._errors and et is None:
            raise self._errors[0]
        return False


def encoder_gaussian_chunk(x, mean, scale, Q, file_name='tmp.b', chunk_size=1000_0000):
    """Pieces of at most chunk_size elements, one `<name>_<c>.b` file each (:317-337)."""
    assert file_name.endswith('.b')
    assert len(x.shape) == 1
    xv, mv, sv = x.view(-1), mean.view(-1), scale.view(-1)
    qv = Q.view(-1) if isinstance(Q, torch.Tensor) else None
    return sum(encoder_gaussian(xv[sl], mv[sl], sv[sl], qv[sl] if qv is not None else Q, file_name.replace('.b', f'_{str(c)}.b'))
               for c, sl in enumerate(_chunk_slices(xv.shape[0], chunk_size)))


def encoder_gaussian(x, mean, scale, Q, file_name='tmp.b'):
    assert file_name.endswith('.b')
    assert len(x.shape) == 1
    Q = _as_q(Q, mean)
    min_value, max_value, data, cnt = arithmetic.encode_gaussian(x.contiguous(), mean.contiguous(), scale.contiguous(), Q.contiguous(), chunk_size_cuda)
    return _write_b(file_name, min_value, max_value, data, cnt)


def decoder_gaussian_chunk(mean, scale, Q, file_name='tmp.b', chunk_size=1000_0000):
    assert file_name.endswith('.b')
    mv, sv = mean.view(-1), scale.view(-1)
    qv = Q.view(-1) if isinstance(Q, torch.Tensor) else None
    out = [decoder_gaussian(mv[sl], sv[sl], qv[sl] if qv is not None else Q, file_name.replace('.b', f'_{str(c)}.b'))
           for c, sl in enumerate(_chunk_slices(mv.shape[0], chunk_size))]
    return torch.cat(out, dim=0).type_as(mean)


def decoder_gaussian(mean, scale, Q, file_name='tmp.b'):
    assert file_name.endswith('.b')
    assert len(mean.shape) == 1
    assert mean.shape == scale.shape
    Q = _as_q(Q, mean)
    min_value, max_value, data, cnt = _read_b(file_name)
    return arithmetic.decode_gaussian(mean.contiguous(), scale.contiguous(), Q.contiguous(), min_value, max_value, data, cnt, chunk_size_cuda)


def encoder(x, file_name='tmp.b'):
    """Bernoulli coder of the hash tables and the masks (HAC/utils/encodings_cuda.py:228-245): every symbol has the row (0, 1 - p, 1), so no (n, 3)
    table is built (120 MB for a million anchors' mask bits) and the payload stays on the host -- the same bytes in the same file."""
    assert file_name[-2:] == '.b'
    x = x.detach().view(-1)
    prob_1 = x.sum() / x.numel()
    p_u = float((1 - prob_1.to(torch.float32)).item())             # the table's middle entry, as the reference computes it (float32)
    sym = torch.floor(x).to(torch.int16)
    data, cnt = arithmetic.encode_const_row(sym.contiguous(), (0.0, p_u, 1.0), chunk_size_cuda)
    cnt_bytes = cnt.tobytes()
    byte_stream_bytes = data.tobytes()
    with open(file_name, 'wb') as fout:
        fout.write(prob_1.to(torch.float32).cpu().numpy().tobytes())
        fout.write(np.array([len(cnt_bytes)]).astype(np.int32).tobytes())
        fout.write(cnt_bytes)
        fout.write(byte_stream_bytes)
    return (len(byte_stream_bytes) + len(cnt_bytes)) * 8 + 32 * 2


def decoder(N_len, file_name='tmp.b', device='cuda'):
    assert file_name[-2:] == '.b'
    with open(file_name, 'rb') as fin:
        prob_1 = torch.tensor(np.frombuffer(fin.read(4), dtype=np.float32).copy())
        len_cnt_bytes = np.frombuffer(fin.read(4), dtype=np.int32)[0]
        cnt = np.frombuffer(fin.read(len_cnt_bytes), dtype=np.int32)
        data = np.frombuffer(fin.read(), dtype=np.uint8)
    p_u = float((1 - prob_1.to(torch.float32)).item())
    return arithmetic.decode_const_row((0.0, p_u, 1.0), data, cnt, chunk_size_cuda, N_len, torch.device(device))


# ---------------------------------------------------------------- all slices of an attribute at once
def encoder_gaussian_slices(x, mean, scale, Q, slice_start, file_names, chunk_size=1000_0000):
    """encoder_gaussian_chunk for MANY slices in one device call: slice s = elements [slice_start[s], slice_start[s+1]) goes
    to file_names[s] (a `*.b` name; as in encoder_gaussian_chunk the file written is `*_0.b`) with its own min / max.
    Same files as calling encoder_gaussian_chunk slice by slice.  Empty slices write nothing.  Returns the bit count per slice."""
    ss = np.asarray(slice_start, dtype=np.int64)
    lens = np.diff(ss)
    assert lens.max(initial=0) <= chunk_size, "slices longer than chunk_size are split by encoder_gaussian_chunk; use it for those"
    keep = np.nonzero(lens > 0)[0]
    bits = [0] * len(lens)
    if keep.size == 0:
        return bits
    sel = torch.cat([torch.arange(int(ss[i]), int(ss[i + 1]), device=x.device) for i in keep]) if keep.size != len(lens) else None
    pick = (lambda t: t.contiguous()) if sel is None else (lambda t: t[sel].contiguous())
    cs = np.concatenate([[0], np.cumsum(lens[keep])])
    mins, maxs, data, cnt = arithmetic.encode_gaussian_slices(pick(x), pick(mean), pick(scale), pick(Q), cs, chunk_size_cuda)
    nch = [-(-int(l) // chunk_size_cuda) for l in lens[keep]]
    c0 = b0 = 0
    jobs = []
    for j, i in enumerate(keep):
        c = cnt[c0:c0 + nch[j]]
        nb = int(c.sum())
        fn = file_names[i].replace('.b', '_0.b')
        blob = b"".join((np.float32(mins[j]).tobytes(), np.float32(maxs[j]).tobytes(), np.array([4 * len(c)]).astype(np.int32).tobytes(),
                         c.tobytes(), data[b0:b0 + nb].tobytes()))
        jobs.append((fn, blob))
        bits[i] = (nb + 4 * len(c)) * 8 + 32 * 3
        c0 += nch[j]; b0 += nb
    _write_files(jobs)
    return bits


def decoder_gaussian_slices(mean, scale, Q, slice_start, file_names):
    """Inverse of encoder_gaussian_slices: reads the per-slice files, decodes every chunk of every slice concurrently.
    Returns the decoded values of all (non-empty) slices concatenated in slice order."""
    ss = np.asarray(slice_start, dtype=np.int64)
    lens = np.diff(ss)
    keep = np.nonzero(lens > 0)[0]
    if keep.size == 0:
        return torch.empty(0, dtype=torch.float32, device=mean.device)
    mins, maxs, cnts, datas = _read_slice_files([file_names[i].replace('.b', '_0.b') for i in keep], lens[keep])
    sel = torch.cat([torch.arange(int(ss[i]), int(ss[i + 1]), device=mean.device) for i in keep]) if keep.size != len(lens) else None
    pick = (lambda t: t.contiguous()) if sel is None else (lambda t: t[sel].contiguous())
    cs = np.concatenate([[0], np.cumsum(lens[keep])])
    return arithmetic.decode_gaussian_slices(pick(mean), pick(scale), pick(Q), cs, np.array(mins), np.array(maxs), np.concatenate(datas),
                                             np.concatenate(cnts), chunk_size_cuda)


def decoder_gaussian_slices_multi(jobs):
    """decoder_gaussian_slices for SEVERAL attributes in one device call: jobs = [(mean, scale, Q, slice_start, file_names), ...]; returns the decoded
    tensor of every job.  The decoder is one wave per 10 000-symbol chunk and bound by the chain of a chunk, not by the GPU (HAC's scaling attribute:
    667 chunks, 12.8 ms; feat: 5 000 chunks, 9.4 ms; offsets 7.2 ms) -- decoded one after the other the three leave most SIMDs idle most of the time;
    as ONE list of slices (each with its own min / max, as in the files) their chunks run side by side and the call takes what its longest chain takes."""
    dev = jobs[0][0].device
    parts, paths, lens_all, sizes = [], [], [], []
    for mean, scale, Q, slice_start, file_names in jobs:
        ss = np.asarray(slice_start, dtype=np.int64)
        lens = np.diff(ss)
        keep = np.nonzero(lens > 0)[0]
        paths += [file_names[i].replace('.b', '_0.b') for i in keep]
        sel = torch.cat([torch.arange(int(ss[i]), int(ss[i + 1]), device=dev) for i in keep]) if (keep.size != len(lens) and keep.size) else None
        pick = (lambda t: t.reshape(-1)) if sel is None else (lambda t, sel=sel: t.reshape(-1)[sel])
        if keep.size:
            parts.append((pick(mean), pick(scale), pick(Q)))
            lens_all.append(lens[keep])
        sizes.append(int(lens[keep].sum()) if keep.size else 0)
    if not parts:
        return [torch.empty(0, dtype=torch.float32, device=dev) for _ in jobs]
    mins, maxs, cnts, datas = _read_slice_files(paths, np.concatenate(lens_all))          # every job's files in one call (the blob is per thread and per call)
    cs = np.concatenate([[0], np.cumsum(np.concatenate(lens_all))])
    cat = lambda k: torch.cat([p[k].float() for p in parts]).contiguous()
    out = arithmetic.decode_gaussian_slices(cat(0), cat(1), cat(2), cs, np.array(mins), np.array(maxs), np.concatenate(datas), np.concatenate(cnts), chunk_size_cuda)
    return list(torch.split(out, sizes))


def encoder_gaussian_mixed_slices(x, mean_list, scale_list, prob_list, Q, slice_start, file_names, chunk_size=1000_0000):
    """encoder_gaussian_mixed_chunk (HAC-plus/utils/encodings_cuda.py:177-225) for MANY slices in one device call: slice s =
    elements [slice_start[s], slice_start[s+1]) goes to file_names[s] (as in the reference the file written is `*_0.b`) with its
    own min / max.  Same files as calling encoder_gaussian_mixed_chunk slice by slice; no slice may be empty or longer than
    chunk_size.  Returns the bit count per slice."""
    ss = np.asarray(slice_start, dtype=np.int64)
    lens = np.diff(ss)
    assert lens.min(initial=1) > 0 and lens.max(initial=0) <= chunk_size
    cont = lambda lst: [t.contiguous() for t in lst]
    mins, maxs, data, cnt = arithmetic.encode_gaussian_mixed_slices(x.contiguous(), cont(mean_list), cont(scale_list), cont(prob_list), Q.contiguous(), ss - ss[0],
                                                                    chunk_size_cuda)
    bits, jobs = [], []
    c0 = b0 = 0
    for i, ln in enumerate(lens):
        nch = -(-int(ln) // chunk_size_cuda)
        c = cnt[c0:c0 + nch]
        nb = int(c.sum())
        blob = b"".join((np.float32(mins[i]).tobytes(), np.float32(maxs[i]).tobytes(), np.array([4 * len(c)]).astype(np.int32).tobytes(),
                         c.tobytes(), data[b0:b0 + nb].tobytes()))
        jobs.append((file_names[i].replace('.b', '_0.b'), blob))
        bits.append((nb + 4 * len(c)) * 8 + 32 * 3)
        c0 += nch; b0 += nb
    _write_files(jobs)
    return bits


def decoder_gaussian_mixed_slices(mean_list, scale_list, prob_list, Q, slice_start, file_names):
    """Inverse of encoder_gaussian_mixed_slices: every chunk of every slice decoded concurrently; the decoded values of all
    slices concatenated in slice order."""
    ss = np.asarray(slice_start, dtype=np.int64)
    mins, maxs, cnts, datas = _read_slice_files([fn.replace('.b', '_0.b') for fn in file_names], np.diff(ss))
    cont = lambda lst: [t.contiguous() for t in lst]
    return arithmetic.decode_gaussian_mixed_slices(cont(mean_list), cont(scale_list), cont(prob_list), Q.contiguous(), ss - ss[0], np.array(mins), np.array(maxs),
                                                   np.concatenate(datas), np.concatenate(cnts), chunk_size_cuda)


# ---------------------------------------------------------------- `.b` container shared by the Gaussian-family coders
def _write_b(file_name, min_value, max_value, byte_stream_torch, cnt_torch):
    """f32 min | f32 max | i32 len(cnt bytes) | cnt | payload  (HAC/utils/encodings_cuda.py:366-376); returns the bit count"""
    cnt_bytes = cnt_torch.cpu().numpy().tobytes()
    byte_stream_bytes = byte_stream_torch.cpu().numpy().tobytes()
    with open(file_name, 'wb') as fout:
        fout.write(np.float32(min_value).tobytes())
        fout.write(np.float32(max_value).tobytes())
        fout.write(np.array([len(cnt_bytes)]).astype(np.int32).tobytes())
        fout.write(cnt_bytes)
        fout.write(byte_stream_bytes)
    return (len(byte_stream_bytes) + len(cnt_bytes)) * 8 + 32 * 3


def _read_b(file_name):
    with open(file_name, 'rb') as fin:
        min_value = float(np.frombuffer(fin.read(4), dtype=np.float32)[0])
        max_value = float(np.frombuffer(fin.read(4), dtype=np.float32)[0])
        len_cnt_bytes = np.frombuffer(fin.read(4), dtype=np.int32)[0]
        cnt = torch.tensor(np.frombuffer(fin.read(len_cnt_bytes), dtype=np.int32).copy())
        data = torch.tensor(np.frombuffer(fin.read(), dtype=np.uint8).copy())
    return min_value, max_value, data, cnt


def _chunk_slices(n, chunk_size):
    return [slice(c * chunk_size, c * chunk_size + chunk_size) for c in range(int(np.ceil(n / chunk_size)))]


def _as_q(Q, like):
    return Q if isinstance(Q, torch.Tensor) else torch.tensor([Q], dtype=like.dtype, device=like.device).repeat(like.shape[0])


# ---------------------------------------------------------------- HAC++: Gaussian mixture
def encoder_gaussian_mixed_chunk(x, mean_list, scale_list, prob_list, Q, file_name='tmp.b', chunk_size=1000_0000):
    assert file_name.endswith('.b')
    assert len(x.shape) == 1
    x_view = x.view(-1)
    means, scales, probs = ([t.view(-1) for t in lst] for lst in (mean_list, scale_list, prob_list))
    assert x_view.shape[0] == means[0].shape[0] == scales[0].shape[0] == probs[0].shape[0]
    q_view = Q.view(-1) if isinstance(Q, torch.Tensor) else None
    bits = 0
    for c, sl in enumerate(_chunk_slices(x_view.shape[0], chunk_size)):
        bits += encoder_gaussian_mixed(x_view[sl], [m[sl] for m in means], [s[sl] for s in scales], [p[sl] for p in probs],
                                       q_view[sl] if q_view is not None else Q, file_name=file_name.replace('.b', f'_{str(c)}.b'))
    return bits


def encoder_gaussian_mixed(x, mean_list, scale_list, prob_list, Q, file_name='tmp.b'):
    """One device call: quantise, min / max, the mixture's CDF entries inside the coder (no (n, max-min+2) table per
    component as in :210-225).  Same `.b` bytes as the table path (tests/test_gpu_attributes.py)."""
    assert file_name.endswith('.b')
    assert len(x.shape) == 1
    Q = _as_q(Q, x)
    assert x.shape == mean_list[0].shape == scale_list[0].shape == prob_list[0].shape == Q.shape
    cont = lambda lst: [t.contiguous() for t in lst]
    min_value, max_value, data, cnt = arithmetic.encode_gaussian_mixed(x.contiguous(), cont(mean_list), cont(scale_list), cont(prob_list),
                                                                       Q.contiguous(), chunk_size_cuda)
    return _write_b(file_name, min_value, max_value, data, cnt)


def decoder_gaussian_mixed_chunk(mean_list, scale_list, prob_list, Q, file_name='tmp.b', chunk_size=1000_0000):
    assert file_name.endswith('.b')
    means, scales, probs = ([t.view(-1) for t in lst] for lst in (mean_list, scale_list, prob_list))
    q_view = Q.view(-1) if isinstance(Q, torch.Tensor) else None
    out = []
    for c, sl in enumerate(_chunk_slices(means[0].shape[0], chunk_size)):
        out.append(decoder_gaussian_mixed([m[sl] for m in means], [s[sl] for s in scales], [p[sl] for p in probs],
                                          q_view[sl] if q_view is not None else Q, file_name=file_name.replace('.b', f'_{str(c)}.b')))
    return torch.cat(out, dim=0).type_as(mean_list[0])


def decoder_gaussian_mixed(mean_list, scale_list, prob_list, Q, file_name='tmp.b'):
    assert file_name.endswith('.b')
    Q = _as_q(Q, mean_list[0])
    assert mean_list[0].shape == scale_list[0].shape == prob_list[0].shape == Q.shape
    min_value, max_value, data, cnt = _read_b(file_name)
    cont = lambda lst: [t.contiguous() for t in lst]
    return arithmetic.decode_gaussian_mixed(cont(mean_list), cont(scale_list), cont(prob_list), Q.contiguous(), min_value, max_value, data, cnt,
                                            chunk_size_cuda)


# ---------------------------------------------------------------- factorized density (a per-channel learned CDF)
def _factorized_table(lower_func, Q, min_value, max_value, dim, rows, device):
    """(rows * dim, max - min + 2) CDF table of :91-105 / :156-165: per channel the pmf |sigmoid(s u) - sigmoid(s l)| between the
    half-integer bounds of every level, accumulated, a leading zero, clamped; the same row for every element of a channel."""
    levels = torch.arange(int(min_value), int(max_value) + 1, dtype=torch.float, device=device)
    samples = levels.view(1, 1, -1).repeat(dim, 1, 1)                       # [C, 1, L]
    lower = lower_func((samples - 0.5) * Q, stop_gradient=False)
    upper = lower_func((samples + 0.5) * Q, stop_gradient=False)
    sign = -torch.sign(torch.add(lower, upper)).detach()
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    cdf = torch.cumsum(pmf, dim=-1)
    table = torch.cat([torch.zeros_like(cdf[..., 0:1]), cdf], dim=-1)       # [C, 1, L + 1]
    table = table.permute(1, 0, 2).contiguous().repeat(rows, 1, 1).view(rows * dim, -1)
    return torch.clamp(table, min=0.0, max=1.0)


def encoder_factorized_chunk(x, lower_func, Q: float = 1, file_name='tmp.b', chunk_size=1000_0000):
    assert file_name.endswith('.b')
    assert len(x.shape) == 2
    return sum(encoder_factorized(x[sl], lower_func, Q, file_name.replace('.b', f'_{str(c)}.b')) for c, sl in enumerate(_chunk_slices(x.shape[0], chunk_size)))


def encoder_factorized(x, lower_func, Q: float = 1, file_name='tmp.b'):
    """x (N, C); lower_func = the entropy model's `_logits_cumulative` (callable on a [C, 1, L] tensor)."""
    assert file_name.endswith('.b')
    assert len(x.shape) == 2
    x_int_round = torch.round(x / Q)
    max_value, min_value = x_int_round.max(), x_int_round.min()
    table = _factorized_table(lower_func, Q, min_value.item(), max_value.item(), x.shape[-1], x.shape[0], x.device)
    sym = (x_int_round - min_value).to(torch.int16).view(-1)
    data, cnt = arithmetic.arithmetic_encode(sym.contiguous(), table.contiguous(), chunk_size_cuda, int(table.shape[0]), int(table.shape[1]))
    return _write_b(file_name, min_value.item(), max_value.item(), data, cnt)


def decoder_factorized_chunk(lower_func, Q, N_len, dim, file_name='tmp.b', device='cuda', chunk_size=1000_0000):
    assert file_name.endswith('.b')
    out = [decoder_factorized(lower_func, Q, min(chunk_size, N_len - c * chunk_size), dim, file_name.replace('.b', f'_{str(c)}.b'), device)
           for c in range(int(np.ceil(N_len / chunk_size)))]
    return torch.cat(out, dim=0)


def decoder_factorized(lower_func, Q, N_len, dim, file_name='tmp.b', device='cuda'):
    assert file_name.endswith('.b')
    min_value, max_value, data, cnt = _read_b(file_name)
    table = _factorized_table(lower_func, Q, min_value, max_value, dim, N_len, device)
    sym = arithmetic.arithmetic_decode(table.contiguous(), data, cnt, chunk_size_cuda, int(table.shape[0]), int(table.shape[1])).to(device).to(torch.float32)
    return ((sym + min_value) * Q).reshape(N_len, dim)
