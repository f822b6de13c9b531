#!/usr/bin/env python3
"""Developer probe: cProfile of hac_codec.conduct_decoding on a synthetic scene.  Usage: prof_conduct_dec.py [n_anchors]"""
import cProfile
import copy
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
    info, _ = hac_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
    for rep in range(2):
        dec = SyntheticGaussianModel(n, seed=3)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        hac_codec.conduct_decoding(dec, d, info, ckpt_path="synthetic")
        torch.cuda.synchronize()
        pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
