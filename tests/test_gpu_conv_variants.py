"""Every variant of the sparse-convolution kernel against the oracle, bit for bit.

The block height (and with it the kernel: cooperative 16-row kernel, wave-serial asm loop at 32 / 64 / 128 / 255 rows, the
HIP C++ loop) is picked from the level size, so a 10 k-point test cloud only ever meets the small-level kernels.  The
policy is read once per process from the environment; each variant therefore runs in its own interpreter:
GAUSPCC_CONV_R forces the block height, GAUSPCC_CONV_ASM=0 the C++ loop, GAUSPCC_CONV_COOP=0 the wave-serial kernel on
16-row blocks, GAUSPCC_COOP_TALL=0 / GAUSPCC_CONV_SPLIT=0 / GAUSPCC_CONV_PAIR=0 switch the round-3 kernels off one at a time, GAUSPCC_CONV_BALANCE the block-height rule (0: the class height, 1: always equal blocks, default 2: equal
blocks when the last round would be mostly empty)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
assert torch.cuda.is_available()
from tests import gpu_helpers as gh
from oracle import oracle as orc
from gauspcc_amd import runtime
from gauspcc_amd.model import tensor_table
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
k = 5
pts = synthetic_cloud(6000, seed=21)
# one convolution with residual + relu on the sorted leaves
order = gh.sort_zyx(pts)
xyz = pts[order]
rng = np.random.RandomState(3)
f = rng.randn(len(xyz), 32).astype(np.float32); res = rng.randn(len(xyz), 32).astype(np.float32)
w = (rng.randn(k ** 3, 32, 32) * 0.1).astype(np.float32)
out, pairs = gh.conv3d(xyz, f, w, k, res=res, relu=True)
ref = orc.conv(f, orc.nbr(xyz, k), w, res=res, relu=True)
assert np.array_equal(out, ref), "conv3d differs from the oracle"
# the whole codec: device bitstream == oracle bitstream, decoded geometry == oracle's
sd = synthetic_state_dict(32, k)
dm = runtime.Model(sd, 32, k, 0)
om = orc.Model(tensor_table(sd, 32, k), 32, k)
data, st = gh.encode(dm, pts, 10)
assert data == orc.encode(om, pts, chunk_log2=10), "bitstream differs from the oracle"
dec, _, _ = gh.decode(dm, data)
assert np.array_equal(dec, orc.decode(om, data)[0])
print("variant ok", pairs)
"""


@pytest.mark.parametrize("env", [
    {"GAUSPCC_CONV_R": "255"},
    {"GAUSPCC_CONV_R": "128"},
    {"GAUSPCC_CONV_R": "64"},
    {"GAUSPCC_CONV_R": "32"},
    {"GAUSPCC_CONV_R": "16", "GAUSPCC_CONV_COOP": "0"},
    {"GAUSPCC_COOP_TALL": "0"},                              # the cooperative kernel on 16-row blocks only (default: 16 / 32 / 64 by level size)
    {"GAUSPCC_CONV_SPLIT": "0"},                             # no products-over-the-chip kernels on the tiniest levels
    {"GAUSPCC_CONV_PAIR": "0"},                              # one-tile asm loop everywhere
    {"GAUSPCC_CONV_R": "128", "GAUSPCC_CONV_ASM": "0"},
    {"GAUSPCC_CONV_R": "255", "GAUSPCC_CONV_ASM": "0"},
    {"GAUSPCC_CONV_R": "255", "GAUSPCC_CONV_BALANCE": "0"},
    {"GAUSPCC_CONV_R": "255", "GAUSPCC_CONV_BALANCE": "1"},
    {"GAUSPCC_CONV_R": "128", "GAUSPCC_CONV_BALANCE": "1"},
], ids=lambda e: ",".join(f"{k[8:]}={v}" for k, v in e.items()))
def test_conv_kernel_variant_bit_exact(env):
    e = dict(os.environ)
    e.update(env)
    e["GAUSPCC_DEV"] = "1"               # kernel-selection knobs are developer switches: ignored without it (csrc/common.hpp: dev_env_int)
    e.setdefault("GAUSPCC_FUSED", "0")   # these variants are about the block-tile kernels: keep the decoder's small levels on them
    r = subprocess.run([sys.executable, "-c", SNIPPET % ROOT], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("env", [
    {"GAUSPCC_FUSED": "1"},                                  # the default: persistent small-level launches (csrc/fused.hpp)
    {"GAUSPCC_FUSED": "2"},                                  # the same pair plans, every phase its own launch
    {"GAUSPCC_FUSED": "1", "GAUSPCC_FUSED_GRID": "8"},       # one workgroup per XCC
    {"GAUSPCC_FUSED": "1", "GAUSPCC_FUSED_GRID": "37"},      # uneven XCC membership
    {"GAUSPCC_FUSED": "1", "GAUSPCC_FUSED_GRID": "256"},     # every CU
    {"GAUSPCC_FUSED": "1", "GAUSPCC_FUSED_MAX": "700"},      # only the tiniest levels fused: the hand-over to the block-tile path three levels earlier
], ids=lambda e: ",".join(f"{k[8:]}={v}" for k, v in e.items()))
def test_fused_small_levels_variant_bit_exact(env):
    """The decoder's small levels through the pair-plan path (reference chain: HAC/utils/pcc_utils.py:283-372): decoded geometry ==
    the oracle's, whatever the grid of the persistent launch and wherever the hand-over to the block-tile kernels lies."""
    e = dict(os.environ)
    e.update(env)
    e["GAUSPCC_DEV"] = "1"
    r = subprocess.run([sys.executable, "-c", SNIPPET % ROOT], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_rank_derivation_equals_rank_sort():
    """Raster ranks derived from the parent level (octree.hip: level_ranks_from_parent) against the radix sort of the (z, y, x)
    keys (GAUSPCC_RANK_SORT=1 forces the sort on every level) and against the one-launch-per-step path for the small levels
    (GAUSPCC_SMALL_FUSE=0 switches off the single-workgroup expansion + rank kernel, octree.hip: k_small_level): the
    bitstreams and the decoded order of a 150 k-point cloud are identical in all three."""
    snippet = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
from tests import gpu_helpers as gh
from gauspcc_amd import runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
dm = runtime.Model(synthetic_state_dict(32, 3), 32, 3, 0)
pts = synthetic_cloud(150000, seed=31, negative=True)
data, st = gh.encode(dm, pts, 10)
dec, _, _ = gh.decode(dm, data)
import hashlib
print("digest", hashlib.sha256(data).hexdigest(), hashlib.sha256(dec.tobytes()).hexdigest(), max(st.level_nodes[: st.num_levels]))
""" % ROOT
    out = []
    for env in ({}, {"GAUSPCC_RANK_SORT": "1"}, {"GAUSPCC_SMALL_FUSE": "0"}):
        e = dict(os.environ)
        e.update(env)
        e["GAUSPCC_DEV"] = "1"
        r = subprocess.run([sys.executable, "-c", snippet], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("digest")][0])
    assert out[0] == out[1] == out[2] and int(out[0].split()[-1]) > 100_000


def test_workspace_growth_retries_are_transparent():
    """Both calls start from a workspace estimate and grow-and-retry when a cloud needs more (deep, sparse trees).  With the
    estimate scaled down (GAUSPCC_ARENA_SCALE, developer knob) the arena runs out at different points of the two calls --
    tree levels, the tile pool, the rank pass queued on the second stream with its temporaries at the top end, feature
    buffers -- and every retry must leave nothing behind: same bitstream, same decoded order as an unconstrained run."""
    snippet = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
from tests import gpu_helpers as gh
from gauspcc_amd import runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
dm = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
import hashlib
for n, seed in ((120000, 3), (7000, 4), (40000, 5)):
    pts = synthetic_cloud(n, seed=seed)
    data, st = gh.encode(dm, pts, 10)
    dec, _, _ = gh.decode(dm, data)
    print("digest", n, hashlib.sha256(data).hexdigest(), hashlib.sha256(dec.tobytes()).hexdigest())
""" % ROOT
    out = []
    for env in ({}, {"GAUSPCC_ARENA_SCALE": "0.3"}, {"GAUSPCC_ARENA_SCALE": "0.12"}, {"GAUSPCC_ARENA_SCALE": "0.04"}):
        e = dict(os.environ)
        e.update(env)
        e["GAUSPCC_DEV"] = "1"
        r = subprocess.run([sys.executable, "-c", snippet], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("digest")])
    assert len(out[0]) == 3 and out[0] == out[1] == out[2] == out[3]


def test_reference_layout_device_coder_equals_host_coder():
    """chunk_log2 = 0 (the reference's container: one torchac stream per level and stage, pcc_utils.py:174-177): since round 6 the coder of this
    layout runs on the host (csrc/hostcoder.hpp); the one-lane-per-stream device coder stays behind GAUSPCC_V0_DEVICE_CODER=1 as the cross-check.
    Both write the oracle's bytes and decode them to the oracle's points."""
    snippet = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
from tests import gpu_helpers as gh
from gauspcc_amd import runtime
from gauspcc_amd.model import tensor_table
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
from oracle import oracle as orc
sd = synthetic_state_dict(32, 5)
dm = runtime.Model(sd, 32, 5, 0)
om = orc.Model(tensor_table(sd, 32, 5), 32, 5)
for n, seed in ((30000, 3), (700, 4)):
    pts = synthetic_cloud(n, seed=seed)
    data, st = gh.encode(dm, pts, 0)
    assert data == orc.encode(om, pts, chunk_log2=0), "bitstream differs from the oracle"
    dec, _, _ = gh.decode(dm, data)
    assert np.array_equal(dec, orc.decode(om, data)[0])
print("variant ok")
""" % ROOT
    for env in ({}, {"GAUSPCC_V0_DEVICE_CODER": "1"}):
        e = dict(os.environ)
        e.update(env)
        e["GAUSPCC_DEV"] = "1"
        r = subprocess.run([sys.executable, "-c", snippet], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
