#!/usr/bin/env python3
"""Developer probe: device bitstream vs oracle bitstream for one cloud, with the position of the first difference mapped
to (level, stage).  Usage: enc_diff.py [points] [seed] [k]; kernel policy through the GAUSPCC_CONV_* environment."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from gauspcc_amd import runtime
from gauspcc_amd.model import tensor_table
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
from oracle import oracle as orc
from tests import gpu_helpers as gh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 21
k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pts = synthetic_cloud(n, seed=seed)
sd = synthetic_state_dict(32, k)
dm = runtime.Model(sd, 32, k, 0)
om = orc.Model(tensor_table(sd, 32, k), 32, k)
for cl in (0,):
    data, st = gh.encode(dm, pts, cl)
    ref = orc.encode(om, pts, chunk_log2=cl)
    lv = list(st.level_nodes[: st.num_levels])
    print("levels", lv, "bytes", len(data), len(ref), "env", {k_: v for k_, v in os.environ.items() if k_.startswith("GAUSPCC")})
    if data == ref:
        print("IDENTICAL")
        continue
    first = next(i for i, (a, b) in enumerate(zip(data, ref)) if a != b)
    bn = int(np.frombuffer(ref[2:6], np.int32)[0])
    pos = 6 + 13 * bn + 2
    si = 0
    bad = []
    while pos < len(ref):
        ln = int(np.frombuffer(ref[pos:pos + 4], np.uint32)[0])
        ln2 = int(np.frombuffer(data[pos:pos + 4], np.uint32)[0]) if pos + 4 <= len(data) else -1
        same = ln == ln2 and data[pos:pos + 4 + ln] == ref[pos:pos + 4 + ln]
        if not same:
            bad.append((si // 4 + 1, si % 4, ln, ln2))
        if ln != ln2:
            break
        pos += 4 + ln
        si += 1
    print("first differing byte", first, "differing streams (level, stage, len ref, len dev):", bad[:12])
