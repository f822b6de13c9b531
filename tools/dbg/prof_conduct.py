#!/usr/bin/env python3
"""Developer probe: cProfile of hac_codec.conduct_encoding on a synthetic scene.  Usage: prof_conduct.py [n_anchors]"""
import cProfile
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from gauspcc_amd import hac_codec
from gauspcc_amd.synth import SyntheticGaussianModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
enc = SyntheticGaussianModel(n, seed=3)
with tempfile.TemporaryDirectory() as d:
    hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")      # warm-up (model upload, arena)
with tempfile.TemporaryDirectory() as d:
    pr = cProfile.Profile()
    pr.enable()
    hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
