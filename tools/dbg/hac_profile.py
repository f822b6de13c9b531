#!/usr/bin/env python3
"""Developer probe: where the HAC attribute loop's wall time goes -- conduct_encoding / conduct_decoding of a synthetic scene under cProfile
(host-side view: the device work shows up in the calls that wait for it).   python tools/dbg/hac_profile.py [anchors] [enc|dec]"""
import cProfile
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gauspcc_amd import hac_codec  # noqa: E402
from gauspcc_amd.synth import SyntheticGaussianModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "dec"
enc = SyntheticGaussianModel(n, seed=3)
with tempfile.TemporaryDirectory() as d:
    hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
    patched, log = hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
    for rep in range(2):
        dec = SyntheticGaussianModel(64, seed=9)
        for k in ("encoding_xyz", "mlp_grid", "mlp_opacity", "mlp_cov", "mlp_color", "x_bound_min", "x_bound_max", "voxel_size"):
            setattr(dec, k, getattr(enc, k))
        dec._anchor_feat = torch.zeros(1, enc.feat_dim, device="cuda:0")
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        if which == "dec":
            pr.enable()
            hac_codec.conduct_decoding(dec, d, patched, ckpt_path="synthetic")
            torch.cuda.synchronize()
            pr.disable()
        else:
            pr.enable()
            hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
            torch.cuda.synchronize()
            pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
