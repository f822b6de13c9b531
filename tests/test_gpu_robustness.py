"""What round 3's two-scenes-in-flight hunt found, kept as tests.  Both run in their own interpreter: a GPU memory fault
aborts the process it happens in.

* Two scenes in flight on one GPU (own context, stream and host thread each): every encode's bytes and every decode's
  points equal the scene's own solo pass.  The range coder's first row ring of round 3 lived in LDS behind LDS-DMA loads and
  hand-counted waits: wrong symbols in 2-16 % of the decodes as soon as a second scene shared the GPU, none on an idle one
  (HISTORY.md section 4; the rows now sit in registers behind ordinary loads).
* Corrupted containers at 1 M points: error or some cloud, never a fault.  The small clouds of
  test_gpu_parity.py::test_corrupted_containers_never_crash never left mapped memory; at this size unwritten rank arrays and
  an unstaged LDS slot in the tile builder did."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    e = dict(os.environ)
    e.setdefault("GPU_MAX_HW_QUEUES", "8")
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_two_scenes_in_flight_are_bit_identical_to_solo_runs():
    r = _run([os.path.join("tools", "inflight_check.py"), "2", "40"], 900)
    assert r.returncode == 0 and "2 scenes x 40 steps: 0 bad" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_corrupted_million_point_containers_never_fault():
    r = _run([os.path.join("tools", "fuzz_containers.py"), "1000000", "32"], 900)
    assert r.returncode == 0 and "fuzz done" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_side_paths_beside_a_geometry_scene_are_deterministic():
    """conduct_encoding / conduct_decoding (files and decoded tensors hashed), generate_neural_gaussians + the rasteriser and
    the fused Gaussian coder on one thread while another loops 1 M-point geometry encodes + decodes: every iteration equals
    the first, and the geometry thread never fails."""
    r = _run([os.path.join("tools", "inflight_side.py"), "8"], 900)
    assert r.returncode == 0 and "8 iterations, 0 bad" in r.stdout and "steps, 0 bad" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_at_four_million_points():
    """DESIGN quotes a 4 M-point figure (size dependence of the step): the driver's own harness at that size -- levels beyond 2^21
    nodes, 32-bit tile addressing near its first million tiles, a 10 GB workspace -- must round-trip bit-identically and print ONE
    JSON line on a clean stdout (bench.py verifies decoded geometry == input geometry before it prints)."""
    import json

    r = _run(["bench.py", "--points", "4000000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--side-anchors", "0", "--scenes-in-flight", "0",
              "--skip-v0", "--skip-stages"], 900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout)                       # the whole of stdout is the one line
    assert line["roundtrip_bit_identical"] is True and line["config"]["points_per_scene"] == 4_000_000
    assert line["coded_nodes"] > 8_000_000 and line["value"] > 10.0    # (warm: ~24 Mpoints/s; the first call of a process grows a 10 GB workspace)


def test_two_contexts_never_spin_on_each_other(tmp_path):
    """Two contexts with persistent small-level launches on one GPU (csrc/fused.hip: FusedGate): their launches are chained per device,
    so neither ever runs into the bounded spin (which would print the fall-back notice and cost ~2 s)."""
    r = _run([os.path.join("tools", "inflight_check.py"), "2", "12"], 600)
    assert r.returncode == 0 and "0 bad" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "timed out" not in r.stderr, r.stderr[-2000:]


def test_abandoned_persistent_launch_falls_back_and_decodes_correctly():
    """A persistent small-level launch needs all its workgroups (csrc/fused.hip).  GAUSPCC_FUSED_TEST_DESERT=3 makes the third such
    launch of the process lose one: the others run into the bounded spin (~1 s), the sticky time-out word ends the launch, the decoder
    repeats the call on the block-tile kernels, says so once on stderr, and the context stays on that path -- every decode, the one
    that hit the time-out included, returns the encoder's points."""
    snippet = r"""
import sys, time
import numpy as np
sys.path.insert(0, %r)
import torch
from tests import gpu_helpers as gh
from gauspcc_amd import runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
dm = runtime.Model(synthetic_state_dict(32, 5), 32, 5, 0)
pts = synthetic_cloud(30000, seed=77)
data, st = gh.encode(dm, pts, 11)
order = gh.sort_zyx(pts)
for i in range(3):
    t0 = time.perf_counter()
    dec, _, _ = gh.decode(dm, data)
    dt = time.perf_counter() - t0
    assert np.array_equal(np.asarray(dec)[gh.sort_zyx(np.asarray(dec))], pts[order]), i
    print("decode", i, "ok", round(dt, 3))
""" % ROOT
    e = dict(os.environ)
    e["GAUSPCC_FUSED_TEST_DESERT"] = "3"
    e.pop("GAUSPCC_FUSED_QUIET", None)
    r = subprocess.run([sys.executable, "-c", snippet], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.count("ok") == 3, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stderr.count("timed out") == 1, r.stderr[-3000:]


def test_oversize_chunk_is_recoded_with_smaller_chunks():
    """A chunk whose bytes exceed the staged decoder's LDS window cannot be written (the limit is part of the format: include/gauspcc.h);
    gpcc_encode codes the cloud again with chunk_log2 - 1 until every chunk fits.  The real limit needs a model that spends > 8 bits per
    16-ary symbol at chunk_log2 >= 13; GAUSPCC_TEST_CHUNK_BYTES stands in for the window here.  The container says which chunk_log2 was
    used, equals the oracle's container at that value, and decodes."""
    snippet = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
from tests import gpu_helpers as gh
from oracle import oracle as orc
from gauspcc_amd import runtime
from gauspcc_amd.model import tensor_table
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
sd = synthetic_state_dict(32, 5)
dm = runtime.Model(sd, 32, 5, 0)
om = orc.Model(tensor_table(sd, 32, 5), 32, 5)
pts = synthetic_cloud(120000, seed=5)
data, st = gh.encode(dm, pts, 11)
used = data[3]
assert data[0] == 0xFF and data[1] == 0xFF and 7 <= used < 11, used
assert data == orc.encode(om, pts, chunk_log2=used), "container differs from the oracle's at the chunk size the retry settled on"
dec, _, _ = gh.decode(dm, data)
assert np.array_equal(dec, orc.decode(om, data)[0])
print("recoded at chunk_log2", used)
""" % ROOT
    e = dict(os.environ)
    e["GAUSPCC_TEST_CHUNK_BYTES"] = "400"
    r = subprocess.run([sys.executable, "-c", snippet], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "recoded at chunk_log2" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_stalled_lookback_scan_raises_an_error_instead_of_a_trap():
    """A device-wide scan whose predecessor tile never publishes (csrc/primitives.hip: k_scan_lookback; GAUSPCC_SCAN_FAULT=2 makes tile 0 of the
    process's second look-back launch stay silent and shortens the wait) used to end in a trap: the process and every context in it gone.  Now the
    context's sticky error word goes up: the call that synchronises returns GPCC_ERR_HIP once, the state is reset, and the next scan -- and a whole
    encode + decode round trip on the same context -- are correct."""
    snippet = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import torch
from tests import gpu_helpers as gh
from tests.test_gpu_primitives import _scan
from gauspcc_amd import _lib, runtime
from gauspcc_amd.synth import synthetic_cloud, synthetic_state_dict
ctx = runtime.context(torch.device("cuda", 0))
x = np.random.RandomState(1).randint(0, 9, size=300000).astype(np.uint32)
ref = np.concatenate([[0], np.cumsum(x[:-1], dtype=np.uint64)]).astype(np.uint32)
out, total = _scan(x)                                   # launch 1: fine
assert np.array_equal(out, ref) and _lib.lib().gpcc_device_error_check(ctx) == 0
out, total = _scan(x)                                   # launch 2: tile 0 silent -> tile 1 gives up, the launch runs out
rc = _lib.lib().gpcc_device_error_check(ctx)
assert rc == -1, rc                                     # GPCC_ERR_HIP, once
assert not np.array_equal(out, ref)
assert _lib.lib().gpcc_device_error_check(ctx) == 0
out, total = _scan(x)                                   # state was reset
assert np.array_equal(out, ref) and total == int(x.sum())
dm = runtime.Model(synthetic_state_dict(32, 3), 32, 3, 0)
pts = synthetic_cloud(200000, seed=5)
data, st = gh.encode(dm, pts, 10)
dec, _, _ = gh.decode(dm, data)
assert np.array_equal(np.asarray(dec)[gh.sort_zyx(np.asarray(dec))], pts[gh.sort_zyx(pts)])
print("scan fault ok")
""" % ROOT
    e = dict(os.environ)
    e["GAUSPCC_DEV"] = "1"
    e["GAUSPCC_SCAN_FAULT"] = "2"
    r = subprocess.run([sys.executable, "-c", snippet], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "scan fault ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
