#!/usr/bin/env python3
"""Race hunt for the callers either side of the geometry codec: one thread loops encode + decode of a 1 M-point scene (the
load), the other repeats the attribute loop and the RD loop of a synthetic HAC-style scene -- conduct_encoding (files hashed),
conduct_decoding (decoded tensors hashed), generate_neural_gaussians + rasteriser (picture hashed), the fused Gaussian coder
and mlp_grid -- and compares every result with its first.  Usage: inflight_side.py [iterations] [anchors]"""
import ctypes as C
import hashlib
import math
import os
import sys
import tempfile
import threading
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from gauspcc_amd import _lib, arithmetic, hac_codec, runtime
from gauspcc_amd.neural_gaussians import generate_neural_gaussians
from gauspcc_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from gauspcc_amd.synth import SyntheticGaussianModel, synthetic_cloud, synthetic_state_dict

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_anchors = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
dev = torch.device("cuda", 0)
L = _lib.lib()
model = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
stop = threading.Event()
load_bad = []


def load():
    x = torch.tensor(synthetic_cloud(1_000_000, seed=77), device=dev)
    h = C.c_void_p()
    _lib.check(L.gpcc_ctx_create(0, C.byref(h)))
    s = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(s.cuda_stream)
    k = 0
    while not stop.is_set():
        pb, nb, st = C.c_void_p(), C.c_int64(), _lib.Stats()
        rc = L.gpcc_encode(h, model.handle, x.data_ptr(), x.shape[0], 11, runtime.f16_bits(1), C.byref(pb), C.byref(nb), C.byref(st), sp)
        px, nn, pq, st2 = C.c_void_p(), C.c_int64(), C.c_uint16(), _lib.Stats()
        if rc == 0:
            rc = L.gpcc_decode(h, model.handle, pb, nb.value, C.byref(px), C.byref(nn), C.byref(pq), C.byref(st2), sp)
        if rc:
            load_bad.append((k, L.gpcc_last_error().decode(errors="replace")))
        k += 1
    s.synchronize()
    L.gpcc_ctx_destroy(h)
    print(f"load thread: {k} steps, {len(load_bad)} bad", load_bad[:3], flush=True)


def sha(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def side():
    W, H = 1600, 1060
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        enc = SyntheticGaussianModel(n_anchors, seed=3)
        g = torch.Generator(device="cpu").manual_seed(1)
        n = 5_000_000
        mean = (torch.randn(n, generator=g) * 2).to(dev); scale = (torch.rand(n, generator=g) * 3 + 0.05).to(dev)
        q = (torch.rand(n, generator=g) * 0.5 + 0.75).to(dev); x = (mean + torch.randn(n, generator=g).to(dev) * scale).contiguous()

        def one():
            out = {}
            with tempfile.TemporaryDirectory() as d:
                patched, _ = hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
                hh = hashlib.sha1()
                for f in sorted(os.listdir(d)):
                    hh.update(open(os.path.join(d, f), "rb").read())
                out["files"] = hh.hexdigest()
                dec = SyntheticGaussianModel(64, seed=9)
                for k in ("encoding_xyz", "mlp_grid", "mlp_opacity", "mlp_cov", "mlp_color", "x_bound_min", "x_bound_max", "voxel_size"):
                    setattr(dec, k, getattr(enc, k))
                dec._anchor_feat = torch.zeros(1, enc.feat_dim, device=dev)
                hac_codec.conduct_decoding(dec, d, patched, ckpt_path="synthetic")
            out["anchor"] = sha(dec._anchor); out["feat"] = sha(dec._anchor_feat); out["scaling"] = sha(dec._scaling); out["offset"] = sha(dec._offset)
            a = dec._anchor.detach()
            ctr = a.mean(dim=0); ext = float((a.max(dim=0).values - a.min(dim=0).values).max())
            eye = ctr + torch.tensor([0.0, 0.0, -1.4 * ext], device=dev)
            Rt = torch.eye(4, device=dev); Rt[:3, 3] = -eye
            fovx = math.radians(60); fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
            zn, zf = 0.01, 100.0
            P = torch.zeros(4, 4, device=dev)
            P[0, 0] = 1 / math.tan(fovx / 2); P[1, 1] = 1 / math.tan(fovy / 2); P[3, 2] = 1.0; P[2, 2] = zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
            view = Rt.T.contiguous(); full = (view @ P.T).contiguous()
            cam = types.SimpleNamespace(camera_center=eye)
            rast = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2),
                                                                    bg=torch.tensor([0.05, 0.1, 0.15], device=dev), scale_modifier=1.0, viewmatrix=view, projmatrix=full,
                                                                    sh_degree=1, campos=eye, prefiltered=False, debug=False))
            xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, dec, None)
            out["gaussians"] = sha(xyz) + sha(color)[:8] + sha(opacity)[:8]
            img, radii = rast(means3D=xyz, means2D=torch.zeros_like(xyz), shs=None, colors_precomp=color, opacities=opacity, scales=scaling, rotations=rot, cov3D_precomp=None)
            out["image"] = sha(img); out["radii"] = sha(radii)
            mn, mx, b, c = arithmetic.encode_gaussian(x, mean, scale, q, 10000)
            out["gauss_bytes"] = sha(b) + sha(c)[:8]
            out["gauss_dec"] = sha(arithmetic.decode_gaussian(mean, scale, q, mn, mx, b, c, 10000))
            torch.cuda.current_stream().synchronize()
            return out

        ref = one()
        bad = 0
        for k in range(iters):
            r = one()
            diff = [key for key in ref if r[key] != ref[key]]
            if diff:
                bad += 1
                print(f"  iteration {k}: differ: {diff}", flush=True)
        print(f"side paths beside a geometry scene: {iters} iterations, {bad} bad", flush=True)


tl = threading.Thread(target=load)
tl.start()
try:
    side()
finally:
    stop.set()
    tl.join()
