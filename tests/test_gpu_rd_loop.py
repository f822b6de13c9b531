"""BASELINE configs[4] stand-in: the full rate-distortion loop on one synthetic HAC scene, end to end on the device --
conduct_encoding -> conduct_decoding -> prefilter (visible_filter) -> generate_neural_gaussians -> GaussianRasterizer at
1600 x 1060 -> PSNR (src/gs_compress/HAC/train.py:385-480; gaussian_renderer/__init__.py:25-172, 199-225, 250-305).
'truck' itself is not available here (no datasets, no network): the scene is SyntheticGaussianModel at 200 k anchors.
The rasteriser output is checked against the oracle's on the same Gaussians within the north star's 0.01 dB."""
import math
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _camera(torch, model, W, H, dev):
    """A camera on the -z side of the scene looking at its centre, 60 degree FoV; matrices stored transposed as
    HAC/scene/cameras.py:48-57 stores them (getWorld2View2 / getProjectionMatrix)."""
    a = model._anchor.detach()
    ctr = a.mean(dim=0); ext = float((a.max(dim=0).values - a.min(dim=0).values).max())
    eye = ctr + torch.tensor([0.0, 0.0, -1.4 * ext], device=dev)
    Rt = torch.eye(4, device=dev); Rt[:3, 3] = -eye
    fovx = math.radians(60); fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    zn, zf = 0.01, 100.0
    P = torch.zeros(4, 4, device=dev)
    P[0, 0] = 1 / math.tan(fovx / 2); P[1, 1] = 1 / math.tan(fovy / 2); P[3, 2] = 1.0; P[2, 2] = zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
    view = Rt.T.contiguous(); full = (view @ P.T).contiguous()
    return types.SimpleNamespace(camera_center=eye), view, full, math.tan(fovx / 2), math.tan(fovy / 2)


def test_rd_loop_encode_decode_render_psnr(orc, tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    from gauspcc_amd import hac_codec
    from gauspcc_amd.neural_gaussians import generate_neural_gaussians
    from gauspcc_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, psnr
    from gauspcc_amd.synth import SyntheticGaussianModel

    dev = torch.device("cuda", 0)
    W, H = 1600, 1060                                      # the <= 1.6K rule of utils/camera_utils.py:25-34
    enc = SyntheticGaussianModel(200_000, seed=3)
    patched, log = hac_codec.conduct_encoding(enc, str(tmp_path), ckpt_path="synthetic")
    assert patched[1] >= 150_000 and "Encoded sizes in MB" in log
    dec = SyntheticGaussianModel(64, seed=9)               # a fresh model: only networks / hash tables / bounds travel as weights
    for k in ("encoding_xyz", "mlp_grid", "mlp_opacity", "mlp_cov", "mlp_color", "x_bound_min", "x_bound_max", "voxel_size"):
        setattr(dec, k, getattr(enc, k))
    dec._anchor_feat = torch.zeros(1, enc.feat_dim, device=dev)
    hac_codec.conduct_decoding(dec, str(tmp_path), patched, ckpt_path="synthetic")
    assert dec._anchor.shape[0] == patched[1]

    cam, view, full, tx, ty = _camera(torch, dec, W, H, dev)
    bg = torch.tensor([0.05, 0.1, 0.15], device=dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tx, tanfovy=ty, bg=bg, scale_modifier=1.0,
                                                            viewmatrix=view, projmatrix=full, sh_degree=1, campos=cam.camera_center,
                                                            prefiltered=False, debug=False))

    def frame(pc):
        # prefilter_voxel (:250-305): anchors as Gaussians with their first three scales, identity rotations
        rot_id = torch.zeros(pc.get_anchor.shape[0], 4, device=dev); rot_id[:, 0] = 1.0
        radii_pure = rast.visible_filter(means3D=pc.get_anchor, scales=pc.get_scaling[:, :3], rotations=rot_id, cov3D_precomp=None)
        visible = radii_pure > 0
        xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, pc, visible)
        img, radii = rast(means3D=xyz, means2D=torch.zeros_like(xyz), shs=None, colors_precomp=color, opacities=opacity, scales=scaling,
                          rotations=rot, cov3D_precomp=None)
        return img, radii, (xyz, color, opacity, scaling, rot), visible

    img, radii, g, visible = frame(dec)
    xyz, color, opacity, scaling, rot = (t.cpu().numpy() for t in g)
    assert int(visible.sum()) > 100_000 and xyz.shape[0] > 500_000 and int((radii > 0).sum()) > 300_000
    ref, rradii, L = orc.raster_forward(bg.cpu().numpy(), W, H, xyz, color, opacity, scaling, 1.0, rot, view.cpu().numpy(), full.cpu().numpy(), tx, ty)
    assert np.array_equal(radii.cpu().numpy(), rradii)
    assert rast.num_rendered == L and L > 1_000_000
    d = np.abs(img.cpu().numpy() - ref)
    assert d.max() < 1e-4, d.max()
    # PSNR of both renders against a common "ground truth" (the encoder-side scene's picture, :417-419)
    gt, _, _, _ = frame(enc)
    gt = gt.clamp(0, 1)
    p_dev = float(psnr(img.clamp(0, 1), gt).mean())
    p_ref = float(psnr(torch.tensor(ref, device=dev).clamp(0, 1), gt).mean())
    assert abs(p_dev - p_ref) < 0.01, (p_dev, p_ref)
    assert p_dev > 50.0        # the decoded scene renders to the encoder-side picture (equal-depth Gaussians aside)


def test_rd_loop_one_million_anchors(tmp_path):
    """The same loop at 1 M anchors -- the largest stand-in for BASELINE configs[2] / [4] the box holds (the scenes themselves
    are not available) -- through size-independent properties: the `.b` / `.bin` file set of conduct_encoding, the decoded
    anchor SET equal to the encoder-side one (geometry is coded losslessly after voxelisation), and the decoded scene
    rendering to the encoder-side picture."""
    import os

    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    from gauspcc_amd import hac_codec
    from gauspcc_amd.neural_gaussians import generate_neural_gaussians
    from gauspcc_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, psnr
    from gauspcc_amd.synth import SyntheticGaussianModel

    dev = torch.device("cuda", 0)
    W, H = 1600, 1060
    enc = SyntheticGaussianModel(1_000_000, seed=3)
    patched, log = hac_codec.conduct_encoding(enc, str(tmp_path), ckpt_path="synthetic")
    n = patched[1]
    assert n >= 900_000 and "Encoded sizes in MB" in log
    names = sorted(os.listdir(tmp_path))
    nslices = -(-n // 3000)                                  # conduct_encoding codes 3000-anchor slices (gaussian_model.py:1123)
    assert "xyz_pcc.bin" in names and sum(f.startswith("feat_") for f in names) == nslices
    assert sum(f.startswith("scaling_") for f in names) == nslices and sum(f.startswith("offsets_") for f in names) == nslices
    dec = SyntheticGaussianModel(64, seed=9)
    for k in ("encoding_xyz", "mlp_grid", "mlp_opacity", "mlp_cov", "mlp_color", "x_bound_min", "x_bound_max", "voxel_size"):
        setattr(dec, k, getattr(enc, k))
    dec._anchor_feat = torch.zeros(1, enc.feat_dim, device=dev)
    hac_codec.conduct_decoding(dec, str(tmp_path), patched, ckpt_path="synthetic")
    assert dec._anchor.shape[0] == n
    # anchor set: decoded voxels == the encoder's voxels (as sets; the decoder returns them in the codec's order)
    vox = lambda a: torch.unique(torch.round(a / enc.voxel_size).to(torch.int64), dim=0)
    vd, ve = vox(dec._anchor), vox(enc._anchor)
    assert vd.shape[0] == n                               # no two decoded anchors share a voxel
    if n == ve.shape[0]:
        assert torch.equal(vd, ve)
    else:                                                 # anchors masked out by the encoder are not coded: a subset
        both = torch.unique(torch.cat([vd, ve]), dim=0)
        assert both.shape[0] == ve.shape[0]

    cam, view, full, tx, ty = _camera(torch, dec, W, H, dev)
    bg = torch.tensor([0.05, 0.1, 0.15], device=dev)
    rast = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tx, tanfovy=ty, bg=bg, scale_modifier=1.0,
                                                            viewmatrix=view, projmatrix=full, sh_degree=1, campos=cam.camera_center,
                                                            prefiltered=False, debug=False))

    def frame(pc):
        xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, pc, None)
        img, radii = rast(means3D=xyz, means2D=torch.zeros_like(xyz), shs=None, colors_precomp=color, opacities=opacity, scales=scaling,
                          rotations=rot, cov3D_precomp=None)
        return img.clamp(0, 1), xyz.shape[0]

    img_dec, ng = frame(dec)
    img_enc, ng_enc = frame(enc)
    assert ng == ng_enc and ng > 4_000_000
    assert float(psnr(img_dec, img_enc).mean()) >= 50.0
