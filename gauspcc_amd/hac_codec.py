"""GaussianModel.conduct_encoding / conduct_decoding of HAC (src/gs_compress/HAC/scene/gaussian_model.py:1090-1222,
1225-1366; the same loop in HAC-plus / TC-GS / CAT-3DGS) on the gfx950 kernels -- SURVEY.md §8(f) row 1.

Both functions take the model as their first argument and touch it only through the attributes the reference methods
touch, so they bind as methods:

    from gauspcc_amd import hac_codec
    GaussianModel.conduct_encoding = hac_codec.conduct_encoding
    GaussianModel.conduct_decoding = hac_codec.conduct_decoding

Same files under `pre_path_name` (xyz_pcc.bin, feat_{s}_0.b, scaling_{s}_0.b, offsets_{s}_0.b, hash.b, masks.b), same
`[N_full, N, MAX_batch_size]` patched-info contract, same log strings.  What runs underneath:

    anchors     calculate_morton_order + compress / decompress_point_cloud     gauspcc_amd.pcc_utils (gpcc_encode / gpcc_decode)
    context     calc_interp_feat (hash grids) -> mlp_grid                      gsge_forward, gshac_mlp2 (specified fp32 order:
                                                                                encoder and decoder see identical parameters)
    attributes  encoder_gaussian_chunk / decoder_gaussian_chunk                gsac_encode_gaussian / gsac_decode_gaussian
                                                                                (CDF entries evaluated inside the coder)
    bits        encoder / decoder (one global Bernoulli p)                     gsac_encode / gsac_decode

The context MLP runs ONCE over all anchors instead of once per 3000-anchor slice (rows are independent, so the slices of
the result are the per-slice results), and all slices of an attribute are coded in ONE device call
(gsac_encode_gaussian_slices / gsac_decode_gaussian_slices): the slices only decide which elements share a `.b` file and
its min / max.  Per million anchors the reference issues ~1000 coder calls, each a serial chain of 10000-symbol chunks.
"""
import os
import time

import torch

from . import _lib, runtime
from .encodings_cuda import decoder, decoder_gaussian_slices, decoder_gaussian_slices_multi, deferred_writes, encoder, encoder_gaussian_slices
from .pcc_utils import calculate_morton_order, compress_point_cloud, decompress_point_cloud

bit2MB_scale = 8 * 1024 * 1024     # HAC/scene/gaussian_model.py:30
MAX_BATCH_SIZE = 3_000             # :1123
Q_FEAT, Q_SCALING, Q_OFFSETS = 1, 0.001, 0.2   # :1147-1149
USE_CLAMP = True                   # HAC/utils/encodings.py:11 (use_clamp)


def default_ckpt_path():
    """HAC looks for <repo>/GausPcgc/best_model_ue_4stage_conv.pt (gaussian_model.py:1110-1112); here: $GAUSPCGC_CKPT."""
    p = os.environ.get("GAUSPCGC_CKPT")
    if not p:
        raise FileNotFoundError("pass ckpt_path=... or set GAUSPCGC_CKPT to best_model_ue_4stage_conv.pt (the reference ships no checkpoint)")
    return p


def mlp2(x, w1, b1, w2, b2):
    """Linear - ReLU - Linear on the device, specified fp32 order (gshac_mlp2).  x (n, din) -> (n, dout)."""
    x = x.contiguous().float()
    w1, b1, w2, b2 = (t.detach().contiguous().float() for t in (w1, b1, w2, b2))
    n, din = x.shape
    dh, dout = w1.shape[0], w2.shape[0]
    y = torch.empty(n, dout, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().gshac_mlp2(runtime.context(x.device), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                     n, din, dh, dout, y.data_ptr(), runtime.stream_ptr(x.device)))
    return y


def grid_mlp(model, feat_context):
    """model.get_grid_mlp(feat_context).  An nn.Sequential(Linear, ReLU, Linear) -- what HAC builds (:258-262) -- goes through
    gshac_mlp2; anything else is called as it is."""
    m = model.get_grid_mlp
    mods = list(m) if isinstance(m, torch.nn.Sequential) else []
    if len(mods) == 3 and isinstance(mods[0], torch.nn.Linear) and isinstance(mods[1], torch.nn.ReLU) and isinstance(mods[2], torch.nn.Linear):
        return mlp2(feat_context, mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias)
    return m(feat_context)


def ste_multistep(x, Q, input_mean):
    """STE_multistep.forward (HAC/utils/encodings.py:55-67)."""
    if USE_CLAMP:
        x = torch.clamp(x, min=(input_mean - 15_000 * Q).detach(), max=(input_mean + 15_000 * Q).detach())
    return torch.round(x / Q) * Q


def _context(model, anchor):
    """(:1152-1172 / :1281-1302) for ALL anchors at once: the 9 split outputs of mlp_grid, flattened per attribute."""
    feat_dim, n_off = model.feat_dim, model.n_offsets
    out = grid_mlp(model, model.calc_interp_feat(anchor))
    mean, scale, mean_scaling, scale_scaling, mean_offsets, scale_offsets, qa_f, qa_s, qa_o = torch.split(
        out, split_size_or_sections=[feat_dim, feat_dim, 6, 6, 3 * n_off, 3 * n_off, 1, 1, 1], dim=-1)
    ctx = {
        "mean": mean.contiguous(), "scale": torch.clamp(scale.contiguous(), min=1e-9),
        "mean_scaling": mean_scaling.contiguous(), "scale_scaling": torch.clamp(scale_scaling.contiguous(), min=1e-9),
        "mean_offsets": mean_offsets.contiguous(), "scale_offsets": torch.clamp(scale_offsets.contiguous(), min=1e-9),
        # one step size per anchor, repeated over the attribute's channels (:1158-1172)
        "Q_feat": (Q_FEAT * (1 + torch.tanh(qa_f.contiguous()))).repeat(1, feat_dim),
        "Q_scaling": (Q_SCALING * (1 + torch.tanh(qa_s.contiguous()))).repeat(1, 6),
        "Q_offsets": (Q_OFFSETS * (1 + torch.tanh(qa_o.contiguous()))).repeat(1, 3 * n_off),
    }
    return ctx


@torch.no_grad()
def conduct_encoding(self, pre_path_name, ckpt_path=None):
    t_codec = 0
    torch.cuda.synchronize(); t1 = time.time()
    print('Start encoding ...')
    mask_anchor = self.get_mask_anchor
    _anchor = self.get_anchor[mask_anchor]
    _feat = self._anchor_feat[mask_anchor]
    _grid_offsets = self._offset[mask_anchor]
    _scaling = self.get_scaling[mask_anchor]
    _mask = self.get_mask[mask_anchor]

    # AI-PCC (:1106-1114)
    _anchor_int = torch.round(_anchor / self.voxel_size)
    sorted_indices = calculate_morton_order(_anchor_int)
    _anchor_int = _anchor_int[sorted_indices]
    npz_path = os.path.join(pre_path_name, 'xyz_pcc.bin')
    out = compress_point_cloud(_anchor_int, ckpt_path or default_ckpt_path(), npz_path)
    bits_xyz = out['file_size_bits']

    _anchor = _anchor_int * self.voxel_size
    _feat = _feat[sorted_indices]
    _grid_offsets = _grid_offsets[sorted_indices]
    _scaling = _scaling[sorted_indices]
    _mask = _mask[sorted_indices]

    N = _anchor.shape[0]
    steps = (N // MAX_BATCH_SIZE) if (N % MAX_BATCH_SIZE) == 0 else (N // MAX_BATCH_SIZE + 1)
    n_off = self.n_offsets
    c = _context(self, _anchor)
    feat_mean, scaling_mean, offsets_mean = _feat.mean(), _scaling.mean(), _grid_offsets.mean()
    hash_b_name = os.path.join(pre_path_name, 'hash.b')
    masks_b_name = os.path.join(pre_path_name, 'masks.b')
    names = lambda stem: [os.path.join(pre_path_name, f'{stem}.b').replace('.b', f'_{s}.b') for s in range(steps)]
    bounds = [min(s * MAX_BATCH_SIZE, N) for s in range(steps + 1)]

    torch.cuda.synchronize(); t0 = time.time()
    with deferred_writes():      # the 3 x 334 slice files are written while the next attribute is coded; all on disk when the block ends
        Q = c["Q_feat"].reshape(-1)
        feat = ste_multistep(_feat.reshape(-1), Q, feat_mean)
        bit_feat_list = encoder_gaussian_slices(feat, c["mean"].reshape(-1), c["scale"].reshape(-1), Q, [b * self.feat_dim for b in bounds], names('feat'))

        Q = c["Q_scaling"].reshape(-1)
        scaling = ste_multistep(_scaling.reshape(-1), Q, scaling_mean)
        bit_scaling_list = encoder_gaussian_slices(scaling, c["mean_scaling"].reshape(-1), c["scale_scaling"].reshape(-1), Q, [b * 6 for b in bounds], names('scaling'))

        mask = _mask.repeat(1, 1, 3).view(-1, 3 * n_off).view(-1).to(torch.bool)        # [N*K*3]
        Q = c["Q_offsets"].reshape(-1)
        offsets = ste_multistep(_grid_offsets.reshape(-1, 3 * n_off).reshape(-1), Q, offsets_mean)
        # masked elements up to each slice boundary: only the sums at the slice boundaries leave the device (a million-entry .tolist() was 10 ms)
        kept = torch.cumsum(mask.view(N, -1).sum(dim=1), dim=0)
        off_bounds = [0] + kept[torch.tensor([b - 1 for b in bounds[1:]], dtype=torch.long, device=kept.device)].cpu().tolist()   # (dtype: an empty list is float32 otherwise and cannot index)
        bit_offsets_list = encoder_gaussian_slices(offsets[mask], c["mean_offsets"].reshape(-1)[mask], c["scale_offsets"].reshape(-1)[mask], Q[mask],
                                                   off_bounds, names('offsets'))
    torch.cuda.synchronize(); t_codec += time.time() - t0

    bit_anchor = bits_xyz
    bit_feat, bit_scaling, bit_offsets = sum(bit_feat_list), sum(bit_scaling_list), sum(bit_offsets_list)
    hash_embeddings = self.get_encoding_params()  # {-1, 1}
    if self.ste_binary:
        bit_hash = encoder(((hash_embeddings.view(-1) + 1) / 2), file_name=hash_b_name)
    else:
        bit_hash = hash_embeddings.numel() * 32
    bit_masks = encoder(_mask, file_name=masks_b_name)

    torch.cuda.synchronize(); t2 = time.time()
    print('encoding time:', t2 - t1)
    print('codec time:', t_codec)
    mlp_bits = self.get_mlp_size()[0] if hasattr(self, "get_mlp_size") else 0
    log_info = f"\nEncoded sizes in MB: " \
               f"anchor {round(bit_anchor/bit2MB_scale, 4)}, " \
               f"feat {round(bit_feat/bit2MB_scale, 4)}, " \
               f"scaling {round(bit_scaling/bit2MB_scale, 4)}, " \
               f"offsets {round(bit_offsets/bit2MB_scale, 4)}, " \
               f"hash {round(bit_hash/bit2MB_scale, 4)}, " \
               f"masks {round(bit_masks/bit2MB_scale, 4)}, " \
               f"MLPs {round(mlp_bits/bit2MB_scale, 4)}, " \
               f"Total {round((bit_anchor + bit_feat + bit_scaling + bit_offsets + bit_hash + bit_masks + mlp_bits)/bit2MB_scale, 4)}, " \
               f"EncTime {round(t2 - t1, 4)}"
    return [self._anchor.shape[0], N, MAX_BATCH_SIZE], log_info


@torch.no_grad()
def conduct_decoding(self, pre_path_name, patched_infos, ckpt_path=None):
    torch.cuda.synchronize(); t1 = time.time()
    print('Start decoding ...')
    [N_full, N, max_batch] = patched_infos
    steps = (N // max_batch) if (N % max_batch) == 0 else (N // max_batch + 1)
    n_off = self.n_offsets
    dev = self._anchor_feat.device
    hash_b_name = os.path.join(pre_path_name, 'hash.b')
    masks_b_name = os.path.join(pre_path_name, 'masks.b')

    masks_decoded = decoder(N * n_off, masks_b_name, device=dev).to(torch.float32).view(-1, n_off, 1)   # {0, 1}

    if self.ste_binary:
        N_hash = torch.zeros_like(self.get_encoding_params()).numel()
        hash_embeddings = decoder(N_hash, hash_b_name, device=dev)  # {0, 1}
        hash_embeddings = (hash_embeddings * 2 - 1).to(torch.float32).view(-1, self.n_features_per_level)
        _install_hash(self, hash_embeddings)      # the context below must run on the decoded tables (bit-identical for {-1, 1})

    npz_path = os.path.join(pre_path_name, 'xyz_pcc.bin')
    anchor_decoded = decompress_point_cloud(npz_path, ckpt_path or default_ckpt_path())
    _anchor_int_dec = anchor_decoded['point_cloud'].to(dev)
    sorted_indices = calculate_morton_order(_anchor_int_dec)
    _anchor_int_dec = _anchor_int_dec[sorted_indices]
    anchor_decoded = _anchor_int_dec * self.voxel_size
    N = anchor_decoded.shape[0]

    c = _context(self, anchor_decoded)
    names = lambda stem: [os.path.join(pre_path_name, f'{stem}.b').replace('.b', f'_{s}.b') for s in range(steps)]
    bounds = [min(s * max_batch, N) for s in range(steps + 1)]
    mask = masks_decoded.repeat(1, 1, 3).view(-1, 3 * n_off).view(-1).to(torch.bool)
    # masked elements up to each slice boundary: only the sums at the slice boundaries leave the device (a million-entry .tolist() was 10 ms)
    kept = torch.cumsum(mask.view(N, -1).sum(dim=1), dim=0)
    off_bounds = [0] + kept[torch.tensor([b - 1 for b in bounds[1:]], dtype=torch.long, device=kept.device)].cpu().tolist()   # (dtype: an empty list is float32 otherwise and cannot index)
    mo = c["mean_offsets"].reshape(-1)
    # the three attributes in ONE device call: their chunks side by side (encodings_cuda.decoder_gaussian_slices_multi); the reference decodes
    # slice after slice, attribute after attribute (:1304-1331)
    feat_decoded, scaling_decoded, offs = decoder_gaussian_slices_multi([
        (c["mean"].reshape(-1), c["scale"].reshape(-1), c["Q_feat"].reshape(-1), [b * self.feat_dim for b in bounds], names('feat')),
        (c["mean_scaling"].reshape(-1), c["scale_scaling"].reshape(-1), c["Q_scaling"].reshape(-1), [b * 6 for b in bounds], names('scaling')),
        (mo[mask], c["scale_offsets"].reshape(-1)[mask], c["Q_offsets"].reshape(-1)[mask], off_bounds, names('offsets'))])
    feat_decoded = feat_decoded.view(N, self.feat_dim)
    scaling_decoded = scaling_decoded.view(N, 6)
    offsets_decoded = torch.zeros_like(mo)
    offsets_decoded[mask] = offs
    offsets_decoded = offsets_decoded.view(N, n_off, 3)

    torch.cuda.synchronize(); t2 = time.time()
    print('decoding time:', t2 - t1)

    print('Start replacing parameters with decoded ones...')
    nn = torch.nn
    self._anchor_feat = nn.Parameter(feat_decoded)
    self._offset = nn.Parameter(offsets_decoded)
    self.decoded_version = True                      # (:1336) the accessors stop applying activations / quantisers
    self._anchor = nn.Parameter(anchor_decoded)
    self._scaling = nn.Parameter(scaling_decoded)
    self._mask = nn.Parameter(masks_decoded)
    print('Parameters are successfully replaced by decoded ones!')
    return f"\nDecTime {round(t2 - t1, 4)}"


def _install_hash(self, hash_embeddings):
    """(:1341-1352) put the decoded hash tables back into the encoders."""
    nn = torch.nn
    enc = self.encoding_xyz
    if getattr(self, "use_2D", False):
        len_3D = enc.encoding_xyz.params.shape[0]
        len_2D = enc.encoding_xy.params.shape[0]
        enc.encoding_xyz.params = nn.Parameter(hash_embeddings[0:len_3D])
        enc.encoding_xy.params = nn.Parameter(hash_embeddings[len_3D:len_3D + len_2D])
        enc.encoding_xz.params = nn.Parameter(hash_embeddings[len_3D + len_2D:len_3D + len_2D * 2])
        enc.encoding_yz.params = nn.Parameter(hash_embeddings[len_3D + len_2D * 2:len_3D + len_2D * 3])
    else:
        enc.params = nn.Parameter(hash_embeddings)
