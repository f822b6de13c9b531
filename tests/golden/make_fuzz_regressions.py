"""Regression fixtures for what the sanitizer fuzzer (tools/fuzz_host.cpp, tools/asan_host.sh) found.  Own fixtures: inputs are
containers made by the oracle and then damaged the way the fuzzer damaged them; expected outputs are what a SAFE reader does.

1. unsorted_base: a container whose base level is not in raster order (two base nodes swapped, coordinates and occupancy).
   Round 5's ASan run: orc_nbr computed its row index from `key[j] - lo` assuming sorted keys -> a write before the row
   (oracle/gpcc_oracle.c: fixed by putting the base level in order on read and bounding the offset).  Expected: the cloud
   of the undamaged container (the base level is a set; its order in the file carries no information).
2. duplicate_base: the same container with base node 1 overwritten by base node 0.  Expected: an error.

    python tests/golden/make_fuzz_regressions.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from gauspcc_amd.model import tensor_table
    from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
    from oracle import oracle as orc

    orc.build()
    om = orc.Model(tensor_table(synthetic_state_dict(32, 3), 32, 3), 32, 3)
    pts = synthetic_cloud(900, seed=77, extent_log2=9)
    good = orc.encode(om, pts, chunk_log2=11)
    L = good[6]
    base_at = 8 + 4 * L + 4
    bn = int.from_bytes(good[base_at:base_at + 4], "little")
    assert bn >= 3
    xyz_at, occ_at = base_at + 4, base_at + 4 + 12 * bn
    b = bytearray(good)
    # swap base nodes 0 and bn - 1
    for a in range(12):
        b[xyz_at + a], b[xyz_at + 12 * (bn - 1) + a] = b[xyz_at + 12 * (bn - 1) + a], b[xyz_at + a]
    b[occ_at], b[occ_at + bn - 1] = b[occ_at + bn - 1], b[occ_at]
    unsorted = bytes(b)
    d = bytearray(good)
    d[xyz_at + 12:xyz_at + 24] = d[xyz_at:xyz_at + 12]
    dup = bytes(d)
    dec, _ = orc.decode(om, good)
    np.savez_compressed(os.path.join(HERE, "fuzz_regressions.npz"), points=pts, good=np.frombuffer(good, dtype=np.uint8), unsorted_base=np.frombuffer(unsorted, dtype=np.uint8),
                        duplicate_base=np.frombuffer(dup, dtype=np.uint8), decoded=dec)
    print("wrote fuzz_regressions.npz:", len(good), "bytes,", bn, "base nodes,", len(dec), "points")


if __name__ == "__main__":
    main()
