#!/usr/bin/env python3
"""Developer probe: the RD frame (generate_neural_gaussians + rasteriser forward) of a synthetic HAC-style scene, a few times, with host-side
phase times -- run it under `rocprofv3 --kernel-trace` and feed the trace to tools/rd_frame_timeline.py to see where a frame's time goes.
    python tools/rd_frame_probe.py [anchors] [frames]"""
import math
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gauspcc_amd.neural_gaussians import generate_neural_gaussians  # noqa: E402
from gauspcc_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from gauspcc_amd.synth import SyntheticGaussianModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W, H = 1600, 1060
dev = torch.device("cuda", 0)
enc = SyntheticGaussianModel(n, seed=0, device="cuda:0")
with torch.no_grad():              # as after conduct_decoding: the activated attributes are stored (the un-decoded model recomputes them with torch ops on every frame)
    enc._anchor, enc._scaling, enc._mask = enc.get_anchor.clone(), enc.get_scaling.clone(), enc.get_mask.clone()
enc.decoded_version = True
ctr = enc._anchor.mean(dim=0); ext = float((enc._anchor.max(dim=0).values - enc._anchor.min(dim=0).values).max())
eye = ctr + torch.tensor([0.0, 0.0, -1.4 * ext], device=dev)
Rt = torch.eye(4, device=dev); Rt[:3, 3] = -eye
fovx = math.radians(60); fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
zn, zf = 0.01, 100.0
P = torch.zeros(4, 4, device=dev)
P[0, 0] = 1 / math.tan(fovx / 2); P[1, 1] = 1 / math.tan(fovy / 2); P[3, 2] = 1.0; P[2, 2] = zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
view = Rt.T.contiguous(); full = (view @ P.T).contiguous()
cam = types.SimpleNamespace(camera_center=eye)
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2), bg=torch.zeros(3, device=dev),
                                         scale_modifier=1.0, viewmatrix=view, projmatrix=full, sh_degree=1, campos=eye, prefiltered=False, debug=False)
rast = GaussianRasterizer(settings)
for f in range(frames):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, enc, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    img, radii = rast(means3D=xyz, means2D=None, shs=None, colors_precomp=color, opacities=opacity, scales=scaling, rotations=rot, cov3D_precomp=None)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"frame {f}: generate call {1e3 * (t1 - t0):.3f} ms (+ {1e3 * (t2 - t1):.3f} to drain), rasterise call {1e3 * (t3 - t2):.3f} ms (+ {1e3 * (t4 - t3):.3f}), "
          f"total {1e3 * (t4 - t0):.3f} ms, {xyz.shape[0]} Gaussians, {rast.num_rendered} tile instances", flush=True)
