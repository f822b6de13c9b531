"""generate_neural_gaussians of HAC (src/gs_compress/HAC/gaussian_renderer/__init__.py:25-172) and of HAC++
(src/gs_compress/HAC-plus/gaussian_renderer/__init__.py:25-205) for RD evaluation -- SURVEY.md §8(f) row 2: anchors -> the
Gaussians the rasteriser draws.  The two differ in two places, both handled here: an un-decoded HAC++ model's mlp_grid has the extra
`prob` head (a TEN-way split, :121-123, against HAC's nine, :103-105), and HAC++ applies the binary offset masks AFTER the opacity
test (:188-205) where HAC multiplies them into the opacity before it (:136-137) -- the same set of Gaussians with the same values
(the masks are {0, 1}), which is what gsnn_generate computes.

    from gauspcc_amd.neural_gaussians import generate_neural_gaussians
    xyz, color, opacity, scaling, rot, time_sub = generate_neural_gaussians(viewpoint_camera, pc, visible_mask)

Inference only (`is_training=False`): the training branches add noise and rate terms and need autograd.  The model `pc`
is used through the attributes the reference uses.  When `pc.decoded_version` is false the attributes are first quantised
with the context model's step sizes exactly as the reference does (:103-114); everything after that -- view vectors,
feature bank, the three MLPs, masking, assembly -- is ONE call into libgauspcc (gsnn_generate: two kernels and a scan
instead of ~40 PyTorch kernels and an (n K, 22) concatenate / mask / split).  For a decoded model the visible anchors go in as an
index list: the kernels read those rows of the model's tensors in place instead of five `tensor[visible_mask]` copies.
"""
import ctypes as C
import time

import torch

from . import _lib, runtime
from .hac_codec import Q_FEAT, Q_OFFSETS, Q_SCALING, grid_mlp, ste_multistep


def quant_steps(pc, anchor):
    """Step sizes of the three attributes from the context model, for an un-decoded model (HAC :103-111, HAC++ :121-132): the last three
    columns of mlp_grid's output whatever heads sit in front of them -- nine-way split (HAC: mean, scale, 2 x scaling, 2 x offsets) or
    ten-way (HAC++: + prob).  Returns (Q_feat (N, F), Q_scaling (N, 6), Q_offsets (N, 3 K)); HAC++ views the last as (N, K, 3), the same values."""
    F, K = pc.feat_dim, pc.n_offsets
    out = grid_mlp(pc, pc.calc_interp_feat(anchor))
    nine, ten = 2 * F + 12 + 6 * K + 3, 3 * F + 12 + 6 * K + 3
    if out.shape[1] not in (nine, ten):
        raise ValueError(f"mlp_grid returns {out.shape[1]} columns; HAC has {nine} (feat_dim {F}, {K} offsets), HAC++ {ten}")
    qa_f, qa_s, qa_o = out[:, -3:-2], out[:, -2:-1], out[:, -1:]
    return ((Q_FEAT * (1 + torch.tanh(qa_f.contiguous()))).repeat(1, F), (Q_SCALING * (1 + torch.tanh(qa_s.contiguous()))).repeat(1, 6),
            (Q_OFFSETS * (1 + torch.tanh(qa_o.contiguous()))).repeat(1, 3 * K))


def _linears(seq):
    mods = [m for m in seq if isinstance(m, torch.nn.Linear)]
    if len(mods) != 2:
        raise TypeError("expected nn.Sequential(Linear, ReLU, Linear, ...) as built in HAC/scene/gaussian_model.py:229-256")
    return [t.detach().float().contiguous() for t in (mods[0].weight, mods[0].bias, mods[1].weight, mods[1].bias)]


@torch.no_grad()
def generate_neural_gaussians(viewpoint_camera, pc, visible_mask=None, is_training=False, step=0):
    if is_training:
        raise NotImplementedError("gauspcc_amd.generate_neural_gaussians is the inference path (RD evaluation); training needs autograd")
    time_sub = 0
    f32 = lambda t: t.detach().float().contiguous()
    dev = pc.get_anchor.device
    K, F = pc.n_offsets, pc.feat_dim
    rows = None
    if not pc.decoded_version:      # quantise as the encoder would (:103-114): on the gathered rows, as the reference does
        if visible_mask is None:
            visible_mask = torch.ones(pc.get_anchor.shape[0], dtype=torch.bool, device=dev)
        anchor = pc.get_anchor[visible_mask]
        feat = pc._anchor_feat[visible_mask]
        grid_offsets = pc._offset[visible_mask]
        grid_scaling = pc.get_scaling[visible_mask]
        binary_grid_masks = pc.get_mask[visible_mask]
        torch.cuda.synchronize(); t1 = time.time()
        q_feat, q_scaling, q_offsets = quant_steps(pc, anchor)
        feat = ste_multistep(feat, q_feat, pc._anchor_feat.mean())
        grid_scaling = ste_multistep(grid_scaling, q_scaling, pc.get_scaling.mean())
        grid_offsets = ste_multistep(grid_offsets.reshape(anchor.shape[0], -1), q_offsets, pc._offset.mean()).view_as(grid_offsets)
        torch.cuda.synchronize(); time_sub = time.time() - t1
        n = anchor.shape[0]
    else:
        # decoded model: the kernels read the visible rows of the model's tensors in place (the reference's five `tensor[visible_mask]`
        # gathers, :54-58, are 0.8 GB of copies per million anchors)
        anchor, feat, grid_offsets, grid_scaling, binary_grid_masks = pc.get_anchor, pc._anchor_feat, pc._offset, pc.get_scaling, pc.get_mask
        n = anchor.shape[0]
        if visible_mask is not None:
            rows = torch.nonzero(visible_mask).view(-1).to(torch.int32)
            n = rows.numel()
    tensors = []
    if getattr(pc, "use_feat_bank", False):
        tensors += _linears(pc.get_featurebank_mlp)
    else:
        tensors += [None] * 4
    tensors += _linears(pc.get_opacity_mlp) + _linears(pc.get_cov_mlp) + _linears(pc.get_color_mlp)
    ptrs = (C.c_void_p * 16)(*[None if t is None else t.data_ptr() for t in tensors])
    anchor, feat, grid_offsets, grid_scaling = f32(anchor), f32(feat), f32(grid_offsets), f32(grid_scaling)
    mask = f32(binary_grid_masks).view(-1, K)
    cam = f32(viewpoint_camera.camera_center).view(3)
    xyz = torch.empty(n * K, 3, device=dev); color = torch.empty(n * K, 3, device=dev); opacity = torch.empty(n * K, 1, device=dev)
    scaling = torch.empty(n * K, 3, device=dev); rot = torch.empty(n * K, 4, device=dev)
    m = C.c_int64()
    _lib.check(_lib.lib().gsnn_generate(runtime.context(dev), n, None if rows is None else rows.data_ptr(), F, K, anchor.data_ptr(), feat.data_ptr(),
                                        grid_offsets.data_ptr(), grid_scaling.data_ptr(), mask.data_ptr(), cam.data_ptr(), ptrs, xyz.data_ptr(), color.data_ptr(),
                                        opacity.data_ptr(), scaling.data_ptr(), rot.data_ptr(), C.byref(m), runtime.stream_ptr(dev)))
    m = m.value
    return xyz[:m], color[:m], opacity[:m], scaling[:m], rot[:m], time_sub
