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


def tensor_table(state_dict: dict, channels: int = 32, kernel_size: int = 5, flip_offsets: bool = False):
    """Return the 39 float32 arrays in C-ABI order, shape-checked.

    flip_offsets reverses the kernel-offset axis of every conv kernel; torchsparse's
    weight-slice <-> (dx,dy,dz) map cannot be verified here, so
    a user with an upstream checkpoint can try both.
    """
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
        w = get(key, (K, C, C))
        t.append(np.ascontiguousarray(w[::-1]) if flip_offsets else w)
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

    sd = torch.load(p, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return sd
