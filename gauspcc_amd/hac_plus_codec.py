"""GaussianModel.conduct_encoding / conduct_decoding of HAC++ (src/gs_compress/HAC-plus/scene/gaussian_model.py:1209-1395,
1396-1590) on the gfx950 kernels -- the successor of hac_codec.py (HAC) among SURVEY.md §8(f)'s rows.

    from gauspcc_amd import hac_plus_codec
    GaussianModel.conduct_encoding = hac_plus_codec.conduct_encoding
    GaussianModel.conduct_decoding = hac_plus_codec.conduct_decoding

Same signatures as the reference methods (HAC++ needs no patched-info argument: the anchor count comes from the decoded
geometry), same files under `pre_path_name` -- xyz_pcc.bin, x_bound_min.pkl / x_bound_max.pkl, feat_{s}_{c}_0.b (five
channel groups c per 3000-anchor slice s), scaling_{s}_0.b, offsets_{s}_0.b, hash.b, masks.b -- same log strings.

What HAC++ adds to HAC's loop, and how it runs here:

    feat        five groups of ten channels, each coded under a TWO-component Gaussian mixture: component 0 from mlp_grid
                (mean, scale, prob), component 1 from the channel-context MLP `get_deform_mlp.forward(feat, mean_scale, to_dec=c)` on
                the groups ALREADY coded (:1306-1321, decode :1490-1504) -- an autoregressive chain over the groups.
                Rows are independent, so group c of ALL anchors is one MLP call (gshac_mlp2_act: Linear - LeakyReLU - Linear in
                the specified fp32 order on the matrix pipe: a context MLP, encoder and decoder must get the same bits) and one
                coder call (gsac_encode / decode_gaussian_mixed_slices: every chunk of every slice concurrently, the
                mixture's CDF entries evaluated inside the coder).  Five MLP + five coder calls per scene instead of
                5 x (N / 3000) of each.
    scaling,    the single-Gaussian coder, all slices per call, as in HAC (gsac_encode / decode_gaussian_slices)
    offsets
    anchors     calculate_morton_order + compress / decompress_point_cloud (gpcc_encode / gpcc_decode)
    context     calc_interp_feat (3-D + three 2-D hash grids) -> mlp_grid (gshac_mlp2), once over all anchors
"""
import os
import time

import torch

from .encodings_cuda import (decoder, decoder_gaussian_mixed_slices, decoder_gaussian_slices, deferred_writes, encoder, encoder_gaussian_mixed_slices,
                             encoder_gaussian_slices)
from .hac_codec import _install_hash, bit2MB_scale, default_ckpt_path, grid_mlp, ste_multistep
from .pcc_utils import calculate_morton_order, compress_point_cloud, decompress_point_cloud
from . import _lib, runtime

MAX_BATCH_SIZE = 3_000                         # HAC-plus/scene/gaussian_model.py:42
Q_FEAT, Q_SCALING, Q_OFFSETS = 1, 0.001, 0.2   # :1277-1279
N_GROUPS, GROUP = 5, 10                        # feat_dim = 50 coded as five groups of ten channels (:1305, Channel_CTX_fea)


def mlp2_act(x, w1, b1, w2, b2, slope):
    """Linear - LeakyReLU(slope) - Linear on the device, specified fp32 order (gshac_mlp2_act).  x (n, din) -> (n, dout)."""
    x = x.contiguous().float()
    w1, b1, w2, b2 = (t.detach().contiguous().float() for t in (w1, b1, w2, b2))
    n, din = x.shape
    dh, dout = w1.shape[0], w2.shape[0]
    y = torch.empty(n, dout, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().gshac_mlp2_act(runtime.context(x.device), x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                         n, din, dh, dout, 1, float(slope), y.data_ptr(), runtime.stream_ptr(x.device)))
    return y


def deform_group(model, feat_q, mean_scale, cc):
    """get_deform_mlp.forward(feat_q, mean_scale, to_dec=cc): (mean_adj, scale_adj, prob_adj), each (N, 10).  Channel_CTX_fea's
    MLP_d{cc} = Sequential(Linear(150 + 10 cc, 40), LeakyReLU, Linear(40, 30)) on cat([d0 .. d(cc-1), mean_scale]) (:117-168) goes
    through gshac_mlp2_act; the `tiny` variant's MLPs (inputs d0 .. d(cc-1) only, :170-220) too; anything else is called as it is."""
    m = model.get_deform_mlp
    sub = getattr(m, f"MLP_d{cc}", None)
    mods = list(sub) if isinstance(sub, torch.nn.Sequential) else []
    if len(mods) == 3 and isinstance(mods[0], torch.nn.Linear) and isinstance(mods[1], torch.nn.LeakyReLU) and isinstance(mods[2], torch.nn.Linear) \
            and mods[2].out_features == 3 * GROUP:
        prev = feat_q[:, :GROUP * cc]
        if mods[0].in_features == GROUP * cc + mean_scale.shape[1]:
            x = torch.cat([prev, mean_scale], dim=-1)
        elif mods[0].in_features == GROUP * cc and cc > 0:
            x = prev
        else:
            return m.forward(feat_q, mean_scale, to_dec=cc)
        y = mlp2_act(x, mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias, mods[1].negative_slope)
        return torch.chunk(y, chunks=3, dim=-1)
    return m.forward(feat_q, mean_scale, to_dec=cc)


def _context(model, anchor):
    """(:1283-1298 / :1467-1483) for ALL anchors at once: the ten split outputs of mlp_grid."""
    feat_dim, n_off = model.feat_dim, model.n_offsets
    out = grid_mlp(model, model.calc_interp_feat(anchor))
    mean, scale, prob, mean_scaling, scale_scaling, mean_offsets, scale_offsets, qa_f, qa_s, qa_o = torch.split(
        out, split_size_or_sections=[feat_dim, feat_dim, feat_dim, 6, 6, 3 * n_off, 3 * n_off, 1, 1, 1], dim=-1)
    return {
        "mean": mean.contiguous(), "scale_raw": scale.contiguous(), "prob": prob.contiguous(),
        "mean_scaling": mean_scaling.contiguous(), "scale_scaling": torch.clamp(scale_scaling.contiguous(), min=1e-9),
        "mean_offsets": mean_offsets.contiguous(), "scale_offsets": torch.clamp(scale_offsets.contiguous(), min=1e-9),
        "Q_feat": (Q_FEAT * (1 + torch.tanh(qa_f.contiguous()))).repeat(1, feat_dim),
        "Q_scaling": (Q_SCALING * (1 + torch.tanh(qa_s.contiguous()))).repeat(1, 6),
        "Q_offsets": (Q_OFFSETS * (1 + torch.tanh(qa_o.contiguous()))).repeat(1, 3 * n_off),
    }


def _group_mixture(model, c, feat_q, cc):
    """The two components of group cc for every anchor (:1306-1308, :1313-1316): flattened (N * 10) tensors."""
    mean_scale = torch.cat([c["mean"], c["scale_raw"], c["prob"]], dim=-1)          # (:1302) the UNclamped scale goes into the context
    scale = torch.clamp(c["scale_raw"], min=1e-9)                                     # (:1303)
    mean_adj, scale_adj, prob_adj = deform_group(model, feat_q, mean_scale, cc)
    sl = slice(cc * GROUP, cc * GROUP + GROUP)
    probs = torch.softmax(torch.stack([c["prob"][:, sl], prob_adj], dim=-1), dim=-1)
    flat = lambda t: t.contiguous().view(-1)
    return ([flat(c["mean"][:, sl]), flat(mean_adj)], [flat(scale[:, sl]), flat(scale_adj)], [flat(probs[..., 0]), flat(probs[..., 1])],
            flat(c["Q_feat"][:, sl]))


def _names(pre_path_name, stem, steps):
    return [os.path.join(pre_path_name, f'{stem}.b').replace('.b', f'_{s}.b') for s in range(steps)]


def get_time():
    """(:43-46) a device-wide sync, then the wall clock: the spans of the `Encoded time in s:` / `Decoded time in s:` lines."""
    torch.cuda.synchronize()
    return time.time()


@torch.no_grad()
def conduct_encoding(self, pre_path_name, ckpt_path=None):
    t_codec = 0
    t_total_0 = get_time()
    torch.cuda.synchronize(); t1 = time.time()
    print('Start encoding ...')
    mask_anchor = self.get_mask_anchor.to(torch.bool)[:, 0]  # N
    _anchor = self.get_anchor[mask_anchor]
    _feat = self._anchor_feat[mask_anchor]       # N, 50
    _grid_offsets = self._offset[mask_anchor]    # N, K, 3
    _scaling = self.get_scaling[mask_anchor]     # N, 6
    _mask = self.get_mask[mask_anchor]           # N, K, 1
    N = _anchor.shape[0]

    t_anchor_0 = get_time()     # (:1234-1248) `anchor` = the order + compress_point_cloud: the in-pipeline timing of the geometry codec
    _anchor_int = torch.round(_anchor / self.voxel_size)
    sorted_indices = calculate_morton_order(_anchor_int)
    _anchor_int = _anchor_int[sorted_indices]
    npz_path = os.path.join(pre_path_name, 'xyz_pcc.bin')
    out = compress_point_cloud(_anchor_int, ckpt_path or default_ckpt_path(), npz_path)
    bits_xyz = out['file_size_bits']
    t_anchor = get_time() - t_anchor_0

    _anchor = _anchor_int * self.voxel_size
    _feat = _feat[sorted_indices]
    _grid_offsets = _grid_offsets[sorted_indices]
    _scaling = _scaling[sorted_indices]
    _mask = _mask[sorted_indices]

    torch.save(self.x_bound_min, os.path.join(pre_path_name, 'x_bound_min.pkl'))
    torch.save(self.x_bound_max, os.path.join(pre_path_name, 'x_bound_max.pkl'))

    steps = (N // MAX_BATCH_SIZE) if (N % MAX_BATCH_SIZE) == 0 else (N // MAX_BATCH_SIZE + 1)
    n_off = self.n_offsets
    c = _context(self, _anchor)
    bounds = [min(s * MAX_BATCH_SIZE, N) for s in range(steps + 1)]
    hash_b_name = os.path.join(pre_path_name, 'hash.b')
    masks_b_name = os.path.join(pre_path_name, 'masks.b')

    torch.cuda.synchronize(); t0 = time.time()
    with deferred_writes():      # the 7 x 334 slice files are written while the next attribute is coded; all on disk when the block ends
        # feat: quantised once (:1298-1299), then group by group under the mixture whose second component the channel-context MLP predicts
        t_feature_0 = get_time()
        feat = ste_multistep(_feat, c["Q_feat"], self._anchor_feat.mean())
        bit_feat = 0
        for cc in range(N_GROUPS):
            means, scales, probs, q = _group_mixture(self, c, feat, cc)
            x = feat[:, cc * GROUP:cc * GROUP + GROUP].contiguous().view(-1)
            names = [fn.replace('.b', f'_{cc}.b') for fn in _names(pre_path_name, 'feat', steps)]
            bit_feat += sum(encoder_gaussian_mixed_slices(x, means, scales, probs, q, [b * GROUP for b in bounds], names, chunk_size=50_0000))
        t_feature = get_time() - t_feature_0

        t_scaling_0 = get_time()
        Q = c["Q_scaling"].reshape(-1)
        scaling = ste_multistep(_scaling.reshape(-1), Q, self.get_scaling.mean())
        bit_scaling = sum(encoder_gaussian_slices(scaling, c["mean_scaling"].reshape(-1), c["scale_scaling"].reshape(-1), Q, [b * 6 for b in bounds],
                                                  _names(pre_path_name, 'scaling', steps), chunk_size=10_0000))
        t_scaling = get_time() - t_scaling_0

        t_offset_0 = get_time()
        mask = _mask.repeat(1, 1, 3).view(-1, 3 * n_off).view(-1).to(torch.bool)        # [N*K*3]
        Q = c["Q_offsets"].reshape(-1)
        offsets = ste_multistep(_grid_offsets.reshape(-1, 3 * n_off).reshape(-1), Q, self._offset.mean())
        # masked elements up to each slice boundary: only the sums at the slice boundaries leave the device (a million-entry .tolist() was 10 ms)
        kept = torch.cumsum(mask.view(N, -1).sum(dim=1), dim=0)
        off_bounds = [0] + kept[torch.tensor([b - 1 for b in bounds[1:]], dtype=torch.long, device=kept.device)].cpu().tolist()   # (dtype: an empty list is float32 otherwise and cannot index)
        bit_offsets = sum(encoder_gaussian_slices(offsets[mask], c["mean_offsets"].reshape(-1)[mask], c["scale_offsets"].reshape(-1)[mask], Q[mask], off_bounds,
                                                  _names(pre_path_name, 'offsets', steps), chunk_size=10_0000))
        t_offset = get_time() - t_offset_0
    torch.cuda.synchronize(); t_codec += time.time() - t0

    bit_anchor = bits_xyz
    t_hash_0 = get_time()
    hash_embeddings = self.get_encoding_params()  # {-1, 1}
    if self.ste_binary:
        bit_hash = encoder(((hash_embeddings.view(-1) + 1) / 2), file_name=hash_b_name)
    else:
        bit_hash = hash_embeddings.numel() * 32
    t_hash = get_time() - t_hash_0
    t_mask_0 = get_time()
    bit_masks = encoder(_mask, file_name=masks_b_name)
    t_mask = get_time() - t_mask_0
    t_total = get_time() - t_total_0

    torch.cuda.synchronize(); t2 = time.time()
    print('encoding time:', t2 - t1)
    print('codec time:', t_codec)
    mlp_bits = self.get_mlp_size()[0] if hasattr(self, "get_mlp_size") else 0
    # 32*3*2/bit2MB_scale is for xyz_bound_min and xyz_bound_max (:1373)
    log_info = f"\nEncoded sizes in MB: " \
               f"anchor {round(bit_anchor/bit2MB_scale, 4)}, " \
               f"feat {round(bit_feat/bit2MB_scale, 4)}, " \
               f"scaling {round(bit_scaling/bit2MB_scale, 4)}, " \
               f"offsets {round(bit_offsets/bit2MB_scale, 4)}, " \
               f"hash {round(bit_hash/bit2MB_scale, 4)}, " \
               f"masks {round(bit_masks/bit2MB_scale, 4)}, " \
               f"MLPs {round(mlp_bits/bit2MB_scale, 4)}, " \
               f"Total {round((bit_anchor + bit_feat + bit_scaling + bit_offsets + bit_hash + bit_masks + mlp_bits)/bit2MB_scale + 32*3*2/bit2MB_scale, 4)}, " \
               f"EncTime {round(t2 - t1, 4)}"
    # (:1384-1392) per-component spans, the same seven fields.  The reference codes slice by slice, so its feat / scaling / offsets spans
    # are sums over the slices; here an attribute's slices are one call each, and the slice files of an attribute are still being written
    # (deferred_writes) while the next attribute's span runs -- the spans add up to less than `Total`, which covers the flush as well.
    log_info_time = f"\nEncoded time in s: " \
               f"anchor {round(t_anchor, 4)}, " \
               f"feat {round(t_feature, 4)}, " \
               f"scaling {round(t_scaling, 4)}, " \
               f"offsets {round(t_offset, 4)}, " \
               f"hash {round(t_hash, 4)}, " \
               f"masks {round(t_mask, 4)}, " \
               f"Total {round(t_total, 4)}"
    return log_info + log_info_time


@torch.no_grad()
def conduct_decoding(self, pre_path_name, ckpt_path=None):
    t_total_0 = get_time()
    torch.cuda.synchronize(); t1 = time.time()
    print('Start decoding ...')
    self.x_bound_min = torch.load(os.path.join(pre_path_name, 'x_bound_min.pkl'))
    self.x_bound_max = torch.load(os.path.join(pre_path_name, 'x_bound_max.pkl'))
    n_off = self.n_offsets
    dev = self._anchor_feat.device
    hash_b_name = os.path.join(pre_path_name, 'hash.b')
    masks_b_name = os.path.join(pre_path_name, 'masks.b')

    t_anchor_0 = get_time()     # (:1422-1438) decompress_point_cloud + the order
    npz_path = os.path.join(pre_path_name, 'xyz_pcc.bin')
    anchor_decoded = decompress_point_cloud(npz_path, ckpt_path or default_ckpt_path())
    _anchor_int_dec = anchor_decoded['point_cloud'].to(dev)
    sorted_indices = calculate_morton_order(_anchor_int_dec)
    _anchor_int_dec = _anchor_int_dec[sorted_indices]
    anchor_decoded = _anchor_int_dec * self.voxel_size
    t_anchor = get_time() - t_anchor_0
    N = anchor_decoded.shape[0]
    steps = (N // MAX_BATCH_SIZE) if (N % MAX_BATCH_SIZE) == 0 else (N // MAX_BATCH_SIZE + 1)

    t_mask_0 = get_time()
    masks_decoded = decoder(N * n_off, masks_b_name, device=dev).to(torch.float32).view(-1, n_off, 1)   # {0, 1}
    t_mask = get_time() - t_mask_0
    t_hash_0 = get_time()
    if self.ste_binary:
        N_hash = torch.zeros_like(self.get_encoding_params()).numel()
        hash_embeddings = decoder(N_hash, hash_b_name, device=dev)  # {0, 1}
        hash_embeddings = (hash_embeddings * 2 - 1).to(torch.float32).view(-1, self.n_features_per_level)
        _install_hash(self, hash_embeddings)      # the context below must run on the decoded tables (bit-identical for {-1, 1})
    t_hash = get_time() - t_hash_0

    c = _context(self, anchor_decoded)
    bounds = [min(s * MAX_BATCH_SIZE, N) for s in range(steps + 1)]
    # feat: the autoregressive chain over the five channel groups (:1484-1504) -- group cc of every anchor at once
    t_feature_0 = get_time()
    feat_decoded = torch.zeros(size=[N, self.feat_dim], device=dev, dtype=torch.float32)
    for cc in range(N_GROUPS):
        means, scales, probs, q = _group_mixture(self, c, feat_decoded, cc)
        names = [fn.replace('.b', f'_{cc}.b') for fn in _names(pre_path_name, 'feat', steps)]
        dec = decoder_gaussian_mixed_slices(means, scales, probs, q, [b * GROUP for b in bounds], names)
        feat_decoded[:, cc * GROUP:cc * GROUP + GROUP] = dec.view(N, GROUP)
    t_feature = get_time() - t_feature_0

    t_scaling_0 = get_time()
    scaling_decoded = decoder_gaussian_slices(c["mean_scaling"].reshape(-1), c["scale_scaling"].reshape(-1), c["Q_scaling"].reshape(-1),
                                              [b * 6 for b in bounds], _names(pre_path_name, 'scaling', steps)).view(N, 6)
    t_scaling = get_time() - t_scaling_0
    t_offset_0 = get_time()
    mask = masks_decoded.repeat(1, 1, 3).view(-1, 3 * n_off).view(-1).to(torch.bool)
    # masked elements up to each slice boundary: only the sums at the slice boundaries leave the device (a million-entry .tolist() was 10 ms)
    kept = torch.cumsum(mask.view(N, -1).sum(dim=1), dim=0)
    off_bounds = [0] + kept[torch.tensor([b - 1 for b in bounds[1:]], dtype=torch.long, device=kept.device)].cpu().tolist()   # (dtype: an empty list is float32 otherwise and cannot index)
    mo = c["mean_offsets"].reshape(-1)
    offsets_decoded = torch.zeros_like(mo)
    offsets_decoded[mask] = decoder_gaussian_slices(mo[mask], c["scale_offsets"].reshape(-1)[mask], c["Q_offsets"].reshape(-1)[mask], off_bounds,
                                                    _names(pre_path_name, 'offsets', steps))
    offsets_decoded = offsets_decoded.view(N, n_off, 3)
    t_offset = get_time() - t_offset_0
    t_total = get_time() - t_total_0

    torch.cuda.synchronize(); t2 = time.time()
    print('decoding time:', t2 - t1)

    print('Start replacing parameters with decoded ones...')
    nn = torch.nn
    _mask = torch.zeros(size=[N, n_off + 1, 1], device=dev)      # (:1545) HAC++ keeps one more mask slot per anchor than it codes
    _mask[:N, :n_off] = masks_decoded
    self._anchor_feat = nn.Parameter(feat_decoded)
    self._offset = nn.Parameter(offsets_decoded)
    self.decoded_version = True
    self._anchor = nn.Parameter(anchor_decoded)
    self._scaling = nn.Parameter(scaling_decoded)
    self._mask = nn.Parameter(_mask)
    print('Parameters are successfully replaced by decoded ones!')
    log_info = f"\nDecTime {round(t2 - t1, 4)}"
    log_info_time = f"\nDecoded time in s: " \
                    f"anchor {round(t_anchor, 4)}, " \
                    f"feat {round(t_feature, 4)}, " \
                    f"scaling {round(t_scaling, 4)}, " \
                    f"offsets {round(t_offset, 4)}, " \
                    f"hash {round(t_hash, 4)}, " \
                    f"masks {round(t_mask, 4)}, " \
                    f"Total {round(t_total, 4)}"
    return log_info + log_info_time
