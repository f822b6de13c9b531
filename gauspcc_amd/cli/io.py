"""Point-cloud file I/O of the codec CLIs (reference: src/ai_pcc/GausPcgc/kit/io.py:12-49).

read_points keeps the reference's two readers -- KITTI `.bin` (float32 x,y,z,intensity) and the
line-oriented ASCII reader that skips every line that is not all numbers (which is how it reads
ASCII PLY) -- and adds `.npy` and binary little-endian PLY, which the reference's glob filter
accepts (compress_ue_4stage_conv.py:58) but its reader cannot parse.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _read_ply_binary(path, header_len, props, count):
    dt = np.dtype([(n, "<" + _PLY_TYPES[t]) for t, n in props])
    with open(path, "rb") as f:
        f.seek(header_len)
        v = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
    return np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float64)


def read_points(filedir):
    ext = os.path.splitext(filedir)[-1]
    if ext == ".bin":
        return np.fromfile(filedir, dtype=np.float32).reshape(-1, 4)[:, :3]
    if ext == ".npy":
        return np.load(filedir)[:, :3]
    if ext == ".ply":
        with open(filedir, "rb") as f:
            head = f.read(4096)
        end = head.find(b"end_header\n")
        if end >= 0 and b"format binary_little_endian" in head[:end]:
            props, count, in_vertex = [], 0, False
            for line in head[:end].decode("ascii", "replace").split("\n"):
                w = line.split()
                if w[:2] == ["element", "vertex"]:
                    count, in_vertex = int(w[2]), True
                elif w[:1] == ["element"]:
                    in_vertex = False
                elif w[:1] == ["property"] and in_vertex:
                    if w[1] == "list":
                        raise ValueError(f"{filedir}: list property inside the vertex element")
                    props.append((w[1], w[2]))
            return _read_ply_binary(filedir, end + len(b"end_header\n"), props, count)
    data = []
    with open(filedir) as f:
        for line in f:
            try:
                vals = [float(v) for v in line.split(" ") if v != "\n" and v != ""]
            except ValueError:
                continue
            if vals:
                data.append(vals)
    return np.array(data)[:, 0:3]


def read_point_clouds(file_path_list, workers=8):
    print("Loading point clouds...")
    with ThreadPoolExecutor(max_workers=workers) as p:
        return list(p.map(read_points, file_path_list))


def run_jobs(fn, items, jobs=1, on_progress=None):
    """fn(item) for every item, results in item order (on_progress(results) after each completed item, with None for the
    items still running: the compress CLI rewrites its CSV there).  jobs > 1: that many host threads, each with its own torch stream (and,
    through runtime.context, its own gpcc context): the files of a batch are independent, and two in flight fill what one
    leaves idle on the GPU (HISTORY.md section 7; the HIP runtime needs GPU_MAX_HW_QUEUES >= 3 x jobs for that, which main()
    exports before the first GPU call)."""
    items = list(items)
    if jobs <= 1 or len(items) <= 1:
        out = []
        for it in items:
            out.append(fn(it))
            if on_progress:
                on_progress(out + [None] * (len(items) - len(out)))
        return out
    import threading
    from concurrent.futures import ThreadPoolExecutor

    import torch

    tls = threading.local()

    gpu = torch.cuda.is_available()   # (the CPU self-test of the sharding path runs the same thread pool without streams)

    def call(it):
        if not gpu:
            return fn(it)
        if not hasattr(tls, "stream"):
            tls.stream = torch.cuda.Stream()
        with torch.cuda.stream(tls.stream):
            r = fn(it)
            tls.stream.synchronize()
        return r

    results = [None] * len(items)
    lock = threading.Lock()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        futs = {ex.submit(call, it): i for i, it in enumerate(items)}
        from concurrent.futures import as_completed

        err = None
        for f in as_completed(futs):
            try:
                r = f.result()
            except Exception as e:   # keep the rows of the other jobs; re-raise when all are done
                err = err or e
                continue
            with lock:
                results[futs[f]] = r
                if on_progress:
                    on_progress(list(results))
        if err is not None:
            raise err
    return results


def export_hw_queues(jobs):
    """Before anything initialises HIP: the runtime maps a process's streams onto 4 hardware queues by default, a gpcc
    context uses three streams."""
    import os

    if jobs > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 4 * jobs)))


_WS_PEAK = [0]


def note_workspace(device=None):
    """Record the high-water mark of the library's own device workspace (all live contexts of this process): job threads and
    their contexts are gone by the time the summary line is printed."""
    from .. import runtime

    try:
        d, _ = runtime.workspace_bytes(device)
    except Exception:
        return
    if d > _WS_PEAK[0]:
        _WS_PEAK[0] = d


def workspace_peak():
    return _WS_PEAK[0]
