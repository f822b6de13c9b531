#!/usr/bin/env python3
"""Developer probe: cProfile of hac_plus_codec.conduct_encoding / conduct_decoding on the synthetic HAC++ scene of tools/bench_side_paths.py."""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gauspcc_amd import hac_plus_codec
from gauspcc_amd.synth import SyntheticGaussianModelPlus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
enc = SyntheticGaussianModelPlus(n, seed=3)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    hac_plus_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
    for what in ("enc", "dec"):
        dec = SyntheticGaussianModelPlus(n, seed=3)
        torch.cuda.synchronize()
        pr = cProfile.Profile(); pr.enable()
        if what == "enc":
            hac_plus_codec.conduct_encoding(enc, d, ckpt_path="synthetic")
        else:
            hac_plus_codec.conduct_decoding(dec, d, ckpt_path="synthetic")
        torch.cuda.synchronize(); pr.disable()
        print("=====", what)
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
