#!/usr/bin/env python3
"""Developer probe: wall time of gauspcc_amd.cli.compress / .decompress over a folder of synthetic clouds with --jobs 1 and 2.
Usage: tools/cli_jobs_probe.py [files] [points]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gauspcc_amd.synth import synthetic_cloud

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
with tempfile.TemporaryDirectory() as d:
    src = os.path.join(d, "src")
    os.makedirs(src)
    for i in range(nf):
        np.save(os.path.join(src, f"c{i}.npy"), synthetic_cloud(n, seed=100 + i).astype(np.float32))
    for jobs in (1, 2, 1, 2):
        out, rec, res = (os.path.join(d, f"{k}{jobs}") for k in ("bin", "rec", "res"))
        common = ["--channels", "32", "--kernel_size", "5", "--ckpt", "synthetic:3", "--jobs", str(jobs)]
        t0 = time.perf_counter()
        subprocess.run([sys.executable, "-m", "gauspcc_amd.cli.compress", "--input_glob", src, "--output_folder", out, "--is_data_pre_quantized", "1", "--posQ", "1",
                        "--resultdir", res, "--prefix", "t"] + common, cwd=ROOT, check=True, capture_output=True)
        t1 = time.perf_counter()
        print(f"jobs {jobs}: compress {nf} x {n} points in {t1 - t0:.2f} s (whole process, incl. start-up and file reads)", flush=True)
