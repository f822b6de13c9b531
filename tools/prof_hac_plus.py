#!/usr/bin/env python3
"""Developer probe: where the host time of HAC++'s conduct_encoding / conduct_decoding goes (wall-clock per call of the coders and of the file
writer, no profiler in the way).  Usage: tools/prof_hac_plus.py [anchors]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gauspcc_amd import arithmetic, encodings_cuda, hac_plus_codec
from gauspcc_amd.synth import SyntheticGaussianModelPlus

acc = {}


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(mod, name, g)


for m, n in ((encodings_cuda, "_write_files"), (arithmetic, "encode_gaussian_mixed_slices"), (arithmetic, "encode_gaussian_slices"), (hac_plus_codec, "_group_mixture"),
             (hac_plus_codec, "_context"), (hac_plus_codec, "compress_point_cloud"), (arithmetic, "decode_gaussian_mixed_slices"), (arithmetic, "decode_gaussian_slices"),
             (hac_plus_codec, "decompress_point_cloud")):
    if hasattr(m, n):
        wrap(m, n)
enc = SyntheticGaussianModelPlus(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, seed=3)
with tempfile.TemporaryDirectory() as d:
    hac_plus_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
for rep in range(2):
    acc.clear()
    with tempfile.TemporaryDirectory() as d:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hac_plus_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("ENCODE %.3f s:" % (t1 - t0), {k: round(v, 3) for k, v in acc.items()})
        acc.clear()
        dec = SyntheticGaussianModelPlus(64, seed=9)
        dec.encoding_xyz, dec.mlp_grid, dec.mlp_deform = enc.encoding_xyz, enc.mlp_grid, enc.mlp_deform
        dec._anchor_feat = torch.zeros(1, enc.feat_dim, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hac_plus_codec.conduct_decoding(dec, d, ckpt_path="synthetic")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("DECODE %.3f s:" % (t1 - t0), {k: round(v, 3) for k, v in acc.items()})
