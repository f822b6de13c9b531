"""Forward-only mirror of `diff_gaussian_rasterization` as the reference uses it
(HAC/gaussian_renderer/__init__.py:20, 199-225, 268-303): GaussianRasterizationSettings,
GaussianRasterizer(...)(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
cov3D_precomp) -> (image (3,H,W), radii (P,) int32) and .visible_filter(...) -> radii.

RD evaluation (render -> PSNR) needs no gradients, so there is no backward; colours must be
precomputed (`shs=None` at every call site of the reference).
"""
import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib, runtime


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    @torch.no_grad()
    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        means3D, scales, rotations, cov3D_precomp = _f32(means3D), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
        P = means3D.shape[0]
        radii = torch.empty(P, dtype=torch.int32, device=means3D.device)       # k_preprocess writes every entry
        view, proj = _f32(rs.viewmatrix), _f32(rs.projmatrix)
        _lib.check(_lib.lib().gsr_visible_filter(
            runtime.context(means3D.device), P, int(rs.image_width), int(rs.image_height), means3D.data_ptr(),
            None if scales is None else scales.data_ptr(), float(rs.scale_modifier), None if rotations is None else rotations.data_ptr(),
            None if cov3D_precomp is None else cov3D_precomp.data_ptr(), view.data_ptr(), proj.data_ptr(), float(rs.tanfovx), float(rs.tanfovy),
            int(bool(rs.prefiltered)), radii.data_ptr(), runtime.stream_ptr(means3D.device)))
        return radii

    @torch.no_grad()
    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if shs is not None:
            raise NotImplementedError("gauspcc_amd.rasterizer: SH evaluation is not on the reference's path (shs=None everywhere); pass colors_precomp")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        means3D, colors, opac = _f32(means3D), _f32(colors_precomp), _f32(opacities)
        scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
        P = means3D.shape[0]
        dev = means3D.device
        H, W = int(rs.image_height), int(rs.image_width)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)       # k_preprocess writes every entry
        view, proj, bg = _f32(rs.viewmatrix), _f32(rs.projmatrix), _f32(rs.bg)
        n = C.c_int64()
        _lib.check(_lib.lib().gsr_forward(
            runtime.context(dev), P, bg.data_ptr(), W, H, means3D.data_ptr(), colors.data_ptr(), opac.data_ptr(),
            None if scales is None else scales.data_ptr(), float(rs.scale_modifier), None if rotations is None else rotations.data_ptr(),
            None if cov3D_precomp is None else cov3D_precomp.data_ptr(), view.data_ptr(), proj.data_ptr(), float(rs.tanfovx), float(rs.tanfovy),
            int(bool(rs.prefiltered)), color.data_ptr(), radii.data_ptr(), C.byref(n), runtime.stream_ptr(dev)))
        self.num_rendered = n.value
        return color, radii


def psnr(img1, img2):
    """HAC/utils/image_utils.py:17-19 (per leading-dim PSNR)."""
    mse = (((img1 - img2)) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
