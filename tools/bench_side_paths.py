#!/usr/bin/env python3
"""Measurements of the paths either side of the geometry codec on a synthetic
HAC-style scene: the attribute loop (conduct_encoding / conduct_decoding), the fused Gaussian coder vs the table path,
generate_neural_gaussians and the rasteriser forward -- the RD loop of BASELINE.json configs[4] (encode -> decode ->
render -> PSNR).  One JSON object on stdout.  Usage: tools/bench_side_paths.py [n_anchors] [W] [H]"""
import json
import math
import os
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gauspcc_amd import arithmetic, hac_codec
from gauspcc_amd.neural_gaussians import generate_neural_gaussians
from gauspcc_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, psnr
from gauspcc_amd.synth import SyntheticGaussianModel

def measure(n_anchors=200_000, W=1600, H=1060, coder_symbols=None, mlp_rows=None):
    """The side paths of SURVEY.md section 8(f) on a synthetic HAC-style scene of n_anchors anchors; returns the dict
    bench.py embeds as `side_paths` (tools/bench_side_paths.py prints it for 1 M anchors: profiles/r03_side_paths.json)."""
    dev = torch.device('cuda', torch.cuda.current_device())
    out = {'n_anchors_requested': n_anchors, 'image': [W, H]}



    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, r


    # ---- the Gaussian coder on one attribute of the whole scene: table path vs fused path
    g = torch.Generator(device="cpu").manual_seed(1)
    n = n_anchors * 50
    mean = (torch.randn(n, generator=g) * 2).to(dev); scale = (torch.rand(n, generator=g) * 3 + 0.05).to(dev)
    q = (torch.rand(n, generator=g) * 0.5 + 0.75).to(dev); x = (mean + torch.randn(n, generator=g).to(dev) * scale).contiguous()


    def table_path():
        xi = torch.round(x / q)
        lower = arithmetic.calculate_cdf(mean, scale, q, xi.min(), xi.max())
        return arithmetic.arithmetic_encode((xi - xi.min()).to(torch.int16).contiguous(), lower, 10000, n, int(lower.shape[1])), lower.shape[1]


    t_tab, ((b_tab, c_tab), lp) = timed(table_path, 2)
    t_fus, (mn, mx, b_fus, c_fus) = timed(lambda: arithmetic.encode_gaussian(x, mean, scale, q, 10000), 2)
    t_dec, xd = timed(lambda: arithmetic.decode_gaussian(mean, scale, q, mn, mx, b_fus, c_fus, 10000), 2)
    out["gaussian_coder"] = {"symbols": n, "alphabet": int(lp) - 1, "table_bytes": n * int(lp) * 4, "encode_table_path_ms": round(t_tab * 1e3, 2),
                             "encode_fused_ms": round(t_fus * 1e3, 2), "decode_fused_ms": round(t_dec * 1e3, 2),
                             "fused_bytes_equal_table_bytes": bool(torch.equal(b_tab.cpu(), b_fus.cpu())),
                             "Msymbols_per_s_encode": round(n / t_fus / 1e6, 1), "Msymbols_per_s_decode": round(n / t_dec / 1e6, 1)}
    del mean, scale, q, x, xd

    # ---- the attribute loop and the RD loop
    enc = SyntheticGaussianModel(n_anchors, seed=3)
    out["n_anchors"] = int(enc._anchor.shape[0])
    with tempfile.TemporaryDirectory() as d:      # warm-up: model upload, workspace growth, first-touch of the pinned buffers
        hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
    with tempfile.TemporaryDirectory() as d:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        patched, log = hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        dec = SyntheticGaussianModel(64, seed=9)
        for k in ("encoding_xyz", "mlp_grid", "mlp_opacity", "mlp_cov", "mlp_color", "x_bound_min", "x_bound_max", "voxel_size"):
            setattr(dec, k, getattr(enc, k))
        dec._anchor_feat = torch.zeros(1, enc.feat_dim, device=dev)
        t2 = time.perf_counter()
        hac_codec.conduct_decoding(dec, d, patched, ckpt_path="synthetic")
        torch.cuda.synchronize(); t3 = time.perf_counter()
        size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    out["attribute_loop"] = {"anchors_coded": patched[1], "files_bytes": size, "conduct_encoding_s": round(t1 - t0, 3), "conduct_decoding_s": round(t3 - t2, 3),
                             "log": log.strip()}

    # ---- the same loop of HAC++ (hac_plus_codec.py: five feat groups under the mixture whose second component comes from the channel-context MLP)
    try:
        from gauspcc_amd import hac_plus_codec
        from gauspcc_amd.synth import SyntheticGaussianModelPlus
        encp = SyntheticGaussianModelPlus(n_anchors, seed=3)
        with tempfile.TemporaryDirectory() as d:
            hac_plus_codec.conduct_encoding(encp, d, ckpt_path="synthetic")     # warm-up: model upload, workspace growth, file-system caches
        # three warm repetitions on the default temporary directory (the box's disk) and three on tmpfs: the 2 343 slice files of a
        # 1 M-anchor scene make the figure depend on where they land (one number in DESIGN.md and under profiles/)
        reps = {}
        logp, sizep, decp = "", 0, None
        for where, root in (("disk", None), ("tmpfs", "/dev/shm" if os.path.isdir("/dev/shm") else None)):
            te, td = [], []
            for _ in range(3):
                with tempfile.TemporaryDirectory(dir=root) as d:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    logp = hac_plus_codec.conduct_encoding(encp, d, ckpt_path="synthetic")
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    decp = SyntheticGaussianModelPlus(64, seed=9)
                    decp.encoding_xyz, decp.mlp_grid, decp.mlp_deform = encp.encoding_xyz, encp.mlp_grid, encp.mlp_deform
                    decp._anchor_feat = torch.zeros(1, encp.feat_dim, device=dev)
                    t2 = time.perf_counter()
                    hac_plus_codec.conduct_decoding(decp, d, ckpt_path="synthetic")
                    torch.cuda.synchronize(); t3 = time.perf_counter()
                    sizep = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
                    te.append(t1 - t0); td.append(t3 - t2)
            reps[where] = {"conduct_encoding_s": [round(x, 3) for x in te], "conduct_decoding_s": [round(x, 3) for x in td]}
        out["attribute_loop_hac_plus"] = {"anchors_coded": int(decp._anchor.shape[0]), "files_bytes": sizep,
                                          "conduct_encoding_s": sorted(reps["disk"]["conduct_encoding_s"])[1], "conduct_decoding_s": sorted(reps["disk"]["conduct_decoding_s"])[1],
                                          "repetitions": reps, "log": logp.strip()}
        del encp, decp
    except Exception as e:   # a side figure must not take the line down
        out["attribute_loop_hac_plus"] = {"error": repr(e)}

    # camera on an orbit around the scene, looking at its centre (HAC/scene/cameras.py conventions: transposed matrices)
    ctr = enc._anchor.mean(dim=0); ext = float((enc._anchor.max(dim=0).values - enc._anchor.min(dim=0).values).max())
    eye = ctr + torch.tensor([0.0, 0.0, -1.4 * ext], device=dev)
    Rt = torch.eye(4, device=dev); Rt[:3, 3] = -eye       # world -> view: translate (camera looks down +z)
    fovx = math.radians(60); fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    zn, zf = 0.01, 100.0
    P = torch.zeros(4, 4, device=dev)
    P[0, 0] = 1 / math.tan(fovx / 2); P[1, 1] = 1 / math.tan(fovy / 2); P[3, 2] = 1.0; P[2, 2] = zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
    view = Rt.T.contiguous(); full = (view @ P.T).contiguous()
    cam = types.SimpleNamespace(camera_center=eye)
    settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(fovx / 2), tanfovy=math.tan(fovy / 2),
                                             bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=view, projmatrix=full, sh_degree=1,
                                             campos=eye, prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)


    def frame(pc):
        xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, pc, None)
        img, radii = rast(means3D=xyz, means2D=torch.zeros_like(xyz), shs=None, colors_precomp=color, opacities=opacity, scales=scaling, rotations=rot,
                          cov3D_precomp=None)
        return img, xyz.shape[0], int((radii > 0).sum())


    t_gen, g_out = timed(lambda: generate_neural_gaussians(cam, dec, None), 3)
    t_frame, (img_dec, n_gauss, n_vis) = timed(lambda: frame(dec), 3)
    img_enc, _, _ = frame(enc)      # the un-decoded model quantises its attributes on the fly: the same picture
    out["rd_loop"] = {"gaussians": n_gauss, "visible": n_vis, "generate_neural_gaussians_ms": round(t_gen * 1e3, 2),
                      "generate_plus_rasterise_ms": round(t_frame * 1e3, 2), "rendered_tiles_instances": int(rast.num_rendered),
                      "psnr_decoded_vs_encoder_side_dB": round(float(psnr(img_dec.clamp(0, 1), img_enc.clamp(0, 1)).mean()), 2),
                      "max_abs_pixel_diff": float((img_dec - img_enc).abs().max())}
    # ---- mlp_grid (HAC's context MLP, 96-100-175) on its own: the matrix-pipe kernel
    rows = mlp_rows or out["n_anchors"]
    gm = torch.Generator(device="cpu").manual_seed(2)
    xm = torch.randn(rows, 96, generator=gm).to(dev)
    w1 = (torch.randn(100, 96, generator=gm) / 10).to(dev); b1 = torch.zeros(100, device=dev)
    w2 = (torch.randn(175, 100, generator=gm) / 10).to(dev); b2 = torch.zeros(175, device=dev)
    t_mlp, _ = timed(lambda: hac_codec.mlp2(xm, w1, b1, w2, b2), 5)
    out["mlp_grid"] = {"rows": rows, "ms": round(t_mlp * 1e3, 3), "TFLOP_per_s": round(rows * 2.0 * (96 * 100 + 100 * 175) / t_mlp / 1e12, 2)}
    del xm
    # ---- the torchac-compatible shim (one stream = one lane: the reference's CPU coder format): float-CDF encode / decode
    from gauspcc_amd import torchac as tac
    nt, lp_t = 200_000, 17
    pt = torch.softmax(torch.randn(nt, lp_t - 1, generator=gm), dim=-1)
    cdf_t = torch.cat([torch.zeros(nt, 1), torch.cumsum(pt, dim=-1)], dim=-1).clamp(0, 1).to(dev)
    sym_t = torch.multinomial(pt, 1).view(-1).to(torch.int16).to(dev)
    t_te, blob = timed(lambda: tac.encode_float_cdf(cdf_t, sym_t), 6)
    t_td, sd = timed(lambda: tac.decode_float_cdf(cdf_t, blob), 6)
    # ... and with CPU tensors, torchac's own calling convention (TC-GS/utils/encodings.py:84-129 moves the table to the CPU first)
    cdf_c, sym_c = cdf_t.cpu(), sym_t.cpu()
    t_ce, blob_c = timed(lambda: tac.encode_float_cdf(cdf_c, sym_c), 6)
    t_cd, _ = timed(lambda: tac.decode_float_cdf(cdf_c, blob_c), 6)
    out["torchac_shim"] = {"symbols": nt, "encode_Msym_per_s": round(nt / t_te / 1e6, 2), "decode_Msym_per_s": round(nt / t_td / 1e6, 2),
                           "cpu_tensors_encode_Msym_per_s": round(nt / t_ce / 1e6, 2), "cpu_tensors_decode_Msym_per_s": round(nt / t_cd / 1e6, 2),
                           "coder": "host thread (csrc/hostcoder.hip): one stream is one dependent chain; int16 rows built on the tensor's device",
                           "roundtrip": bool(torch.equal(sd.to(dev), sym_t)) and blob == blob_c}
    # ... and the reference's ten-way fan-out (TC-GS/utils/encodings.py:36-82: ten chunk files, ten independent streams) on 2 M symbols: ten native threads
    from gauspcc_amd import torchac_encodings as te
    nf = 2_000_000
    pf_ = torch.softmax(torch.randn(nf, lp_t - 1, generator=gm), dim=-1)
    cdf_f = torch.cat([torch.zeros(nf, 1), torch.cumsum(pf_, dim=-1)], dim=-1).clamp(0, 1).to(dev)
    sym_f = torch.multinomial(pf_, 1).view(-1).to(torch.int16).to(dev)
    rows_f = tac._to_int_rows(cdf_f)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        fn = os.path.join(td, "fan.b")
        t_fe, _ = timed(lambda: te.multiprocess_encoder(rows_f, sym_f, fn), 2)
        t_fd, sf = timed(lambda: te.multiprocess_deoder(rows_f, fn), 2)
    out["torchac_shim"].update({"fan_out_symbols": nf, "fan_out_chunks": 10, "fan_out_encode_Msym_per_s": round(nf / t_fe / 1e6, 2),
                                "fan_out_decode_Msym_per_s": round(nf / t_fd / 1e6, 2), "fan_out_roundtrip": bool(torch.equal(sf.to(torch.int16), sym_f))})
    return out


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, int(sys.argv[2]) if len(sys.argv) > 2 else 1600,
                             int(sys.argv[3]) if len(sys.argv) > 3 else 1060)))
