"""Per-device context and model cache above the C ABI (host-side plumbing only:
PyTorch supplies device memory and the current HIP stream)."""
import ctypes as C
import sys
import threading
import weakref

import numpy as np
import torch

from . import _lib
from .model import load_state_dict, tensor_table

_MODELS = {}
_MODELS_LOCK = threading.Lock()
_ALL = weakref.WeakSet()       # every live _Contexts (one per host thread that used the library); weak: a thread's contexts die with it
_ALL_LOCK = threading.Lock()


class _Contexts:
    """The gpcc contexts of one host thread (device index -> handle).  A context serves one call at a time, so every
    thread gets its own: two threads with their own torch streams run their calls concurrently on one GPU (two scenes in
    flight fill what one leaves idle, HISTORY.md section 7).  Destroyed with the thread."""

    def __init__(self):
        self.h = {}
        with _ALL_LOCK:
            _ALL.add(self)

    def __del__(self, _finalizing=sys.is_finalizing):
        try:
            if _finalizing():   # at interpreter exit the HIP runtime may already be shutting down: the process ends anyway
                return
            for h in self.h.values():
                _lib.lib().gpcc_ctx_destroy(h)
        except Exception:
            pass


_TLS = threading.local()


def workspace_bytes(device=None):
    """(device_bytes, pinned_bytes) held by every live context of this process (all host threads), optionally of one device:
    the library's own hipMalloc / hipHostMalloc allocations, which torch.cuda.max_memory_allocated() cannot see."""
    dev = pin = 0
    idx = None if device is None else _device_index(device)
    with _ALL_LOCK:
        owners = list(_ALL)
    for own in owners:
        for i, h in list(own.h.items()):
            if idx is not None and i != idx:
                continue
            d, p = C.c_int64(0), C.c_int64(0)
            _lib.check(_lib.lib().gpcc_ctx_bytes(h, C.byref(d), C.byref(p)))
            dev += d.value
            pin += p.value
    return dev, pin


def _device_index(device=None) -> int:
    if not torch.cuda.is_available():
        raise RuntimeError("gauspcc_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if device is None:
        return torch.cuda.current_device()
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def context(device=None):
    idx = _device_index(device)
    own = getattr(_TLS, "ctx", None)
    if own is None:
        own = _TLS.ctx = _Contexts()
    if idx not in own.h:
        h = C.c_void_p()
        _lib.check(_lib.lib().gpcc_ctx_create(idx, C.byref(h)))
        own.h[idx] = h
    return own.h[idx]


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(_device_index(device)).cuda_stream)


class Model:
    """Weights resident on one device (gpcc_model)."""

    def __init__(self, state_dict, channels=32, kernel_size=5, device=None, flip_offsets=False, offset_order="xyz"):
        self.device = _device_index(device)
        self.channels, self.kernel_size = channels, kernel_size
        tensors = tensor_table(state_dict, channels, kernel_size, flip_offsets, offset_order)
        self._keep = tensors
        ptrs = (C.c_void_p * len(tensors))(*[t.ctypes.data for t in tensors])
        self.handle = C.c_void_p()
        _lib.check(_lib.lib().gpcc_model_create(context(self.device), channels, kernel_size, ptrs, C.byref(self.handle)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().gpcc_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def get_model(ckpt_path, channels=32, kernel_size=5, device=None, flip_offsets=None, offset_order=None) -> Model:
    """The reference rebuilds and reloads the network on every call (pcc_utils.py:65-67);
    here a model is uploaded once per (checkpoint, device, offset layout) and reused.  `ckpt_path` is what the reference's
    functions take -- a torch checkpoint of `net.state_dict()` -- or an .npz / a dict with the same keys / 'synthetic[:seed]'
    / a Model.  The conv-kernel offset layout of the checkpoint (model.conv_offset_layout) comes from the arguments or, for
    callers that only pass a path through compress_point_cloud, from GAUSPCC_OFFSET_ORDER (xyz | zyx) and
    GAUSPCC_FLIP_OFFSETS (0 | 1)."""
    import os

    idx = _device_index(device)
    if isinstance(ckpt_path, Model):
        return ckpt_path
    if offset_order is None:
        offset_order = os.environ.get("GAUSPCC_OFFSET_ORDER", "xyz")
    if flip_offsets is None:
        flip_offsets = os.environ.get("GAUSPCC_FLIP_OFFSETS", "0") not in ("", "0")
    key = (id(ckpt_path) if isinstance(ckpt_path, dict) else str(ckpt_path), channels, kernel_size, idx, bool(flip_offsets), offset_order)
    with _MODELS_LOCK:   # threads share the models (weights are read-only on the device), only contexts are per thread
        if key not in _MODELS:
            _MODELS[key] = Model(load_state_dict(ckpt_path, channels, kernel_size), channels, kernel_size, idx, bool(flip_offsets), offset_order)
        return _MODELS[key]


def f16_bits(v) -> int:
    return int(np.array(v, dtype=np.float16).view(np.uint16))


def bits_f16(b: int) -> np.float16:
    return np.array(b, dtype=np.uint16).view(np.float16)[()]
