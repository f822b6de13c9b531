"""Per-device context and model cache above the C ABI (host-side plumbing only:
PyTorch supplies device memory and the current HIP stream)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .model import load_state_dict, tensor_table

_CTX = {}
_MODELS = {}


def _device_index(device=None) -> int:
    if not torch.cuda.is_available():
        raise RuntimeError("gauspcc_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if device is None:
        return torch.cuda.current_device()
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def context(device=None):
    idx = _device_index(device)
    if idx not in _CTX:
        h = C.c_void_p()
        _lib.check(_lib.lib().gpcc_ctx_create(idx, C.byref(h)))
        _CTX[idx] = h
    return _CTX[idx]


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(_device_index(device)).cuda_stream)


class Model:
    """Weights resident on one device (gpcc_model)."""

    def __init__(self, state_dict, channels=32, kernel_size=5, device=None, flip_offsets=False):
        self.device = _device_index(device)
        self.channels, self.kernel_size = channels, kernel_size
        tensors = tensor_table(state_dict, channels, kernel_size, flip_offsets)
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


def get_model(ckpt_path, channels=32, kernel_size=5, device=None) -> Model:
    """The reference rebuilds and reloads the network on every call (pcc_utils.py:65-67);
    here a model is uploaded once per (checkpoint, device) and reused."""
    idx = _device_index(device)
    if isinstance(ckpt_path, Model):
        return ckpt_path
    key = (id(ckpt_path) if isinstance(ckpt_path, dict) else str(ckpt_path), channels, kernel_size, idx)
    if key not in _MODELS:
        _MODELS[key] = Model(load_state_dict(ckpt_path, channels, kernel_size), channels, kernel_size, idx)
    return _MODELS[key]


def f16_bits(v) -> int:
    return int(np.array(v, dtype=np.float16).view(np.uint16))


def bits_f16(b: int) -> np.float16:
    return np.array(b, dtype=np.uint16).view(np.float16)[()]
