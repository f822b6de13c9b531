"""GausPcgc weights: upstream state-dict -> the flat tensor table of the C ABI.

Mirrors what `Network(channels, kernel_size).load_state_dict(torch.load(ckpt))`
does in the reference (HAC/utils/pcc_utils.py:65-67), minus torchsparse: the key
set is the one listed in SURVEY.md section 2.4 (derived from
network_ue_4stage_conv.py:15-98 and kit/nn.py:14-16,31,106).

Table order == `gpcc_tensor_id` in include/gauspcc.h.
"""
import numpy as np

from .synth import CONV_KEYS, STAGE_M, synthetic_state_dict

T_COUNT = 39


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def conv_offset_layout(w: np.ndarray, kernel_size: int, flip_offsets: bool = False, offset_order: str = "xyz") -> np.ndarray:
    """A (k^3, Cin, Cout) conv kernel as stored by the checkpoint -> this library's enumeration of the k^3 offsets
    (slice o = (dx+r) + k (dy+r) + k^2 (dz+r), x fastest; pinned against a dense F.conv3d in tests/test_oracle_independent.py).

    torchsparse's weight-slice <-> (dx,dy,dz) map cannot be verified here (the package is absent), so the
    two ambiguities a user with an upstream checkpoint may meet are loader options:
      offset_order="zyx"  the checkpoint enumerates z fastest (slice = (dz+r) + k (dy+r) + k^2 (dx+r)): a (k,k,k) axis
                          transpose (x <-> z);
      flip_offsets=True   the checkpoint's slice o belongs to the offset -delta (out[i] += in[i - delta] @ W[o]): the
                          reversed enumeration (equivalent to mirroring all three axes)."""
    k = kernel_size
    if offset_order not in ("xyz", "zyx"):
        raise ValueError("offset_order must be 'xyz' or 'zyx'")
    if offset_order == "zyx":
        w = w.reshape(k, k, k, *w.shape[1:]).transpose(2, 1, 0, 3, 4).reshape(w.shape)
    if flip_offsets:
        w = w[::-1]
    return np.ascontiguousarray(w)


def tensor_table(state_dict: dict, channels: int = 32, kernel_size: int = 5, flip_offsets: bool = False, offset_order: str = "xyz"):
    """Return the 39 float32 arrays in C-ABI order, shape-checked.  flip_offsets / offset_order: conv_offset_layout."""
    C, K = channels, kernel_size ** 3
    sd = {k[7:] if k.startswith("module.") else k: v for k, v in state_dict.items()}

    def get(key, shape):
        if key not in sd:
            raise KeyError(f"checkpoint is missing '{key}'")
        a = _np(sd[key])
        if a.size != int(np.prod(shape)):
            raise ValueError(f"'{key}' has shape {a.shape}, expected {shape}")
        return np.ascontiguousarray(a.reshape(shape))

    t = [get("prior_embedding.weight", (256, C))]
    for key in CONV_KEYS:
        t.append(conv_offset_layout(get(key, (K, C, C)), kernel_size, flip_offsets, offset_order))
    t.append(get("target_embedding.target_res_embedding.weight", (8, C)))
    t += [get(f"pred_head_s{s}.0.weight", (C, C)) for s in range(4)]
    t += [get(f"pred_head_s{s}.0.bias", (C,)) for s in range(4)]
    t += [get(f"pred_head_s{s}.2.weight", (STAGE_M[s], C)) for s in range(4)]
    t += [get(f"pred_head_s{s}.2.bias", (STAGE_M[s],)) for s in range(4)]
    t += [get(f"pred_head_s{s}_emb.weight", ((2, 4, 16)[s - 1], C)) for s in (1, 2, 3)]
    assert len(t) == T_COUNT
    return t


def load_state_dict(ckpt_path, channels: int = 32, kernel_size: int = 5):
    """`ckpt_path` is a torch checkpoint with the upstream keys, an .npz with the
    same keys, or the string 'synthetic[:seed]' (no checkpoint ships with the
    reference: README.md:73-77)."""
    if isinstance(ckpt_path, dict):
        return ckpt_path
    p = str(ckpt_path)
    if p.startswith("synthetic"):
        seed = int(p.split(":")[1]) if ":" in p else 7
        return synthetic_state_dict(channels, kernel_size, seed=seed)
    if p.endswith(".npz"):
        with np.load(p) as z:
            return {k: z[k] for k in z.files}
    import torch

    # upstream saves `net.state_dict()` (src/ai_pcc/GausPcgc/train.py:212-228) and loads it with
    # torch.load(ckpt_path, map_location=device) (pcc_utils.py:66, 267); tensors only, so weights_only loading suffices
    # (a checkpoint that carries other pickled objects is refused: unpickling it would run its code.  A user who
    # trusts such a file opts in with GAUSPCC_UNSAFE_CKPT=1, which is exactly what upstream's plain torch.load does.)
    import os
    import pickle

    try:
        sd = torch.load(p, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if os.environ.get("GAUSPCC_UNSAFE_CKPT") != "1":
            raise ValueError(
                f"{p}: not a tensors-only checkpoint ({e}); set GAUSPCC_UNSAFE_CKPT=1 to unpickle it anyway "
                "(this executes code stored in the file)") from e
        sd = torch.load(p, map_location="cpu", weights_only=False)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if not isinstance(sd, dict):
        raise ValueError(f"{p}: expected a state dict, got {type(sd).__name__}")
    return sd
