"""Multi-GPU readiness: the N > 1 path on real RCCL whenever the GPU box has two devices.  On a 1-GPU
lease the tests skip; the first box with two GPUs turns them into evidence.  Partitioning: independent scenes shard across the
ranks, no data-path collective, one all_gather of per-scene records and a MAX all-reduce of the elapsed time."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_gpus():
    import torch

    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def _run_bench(extra):
    # a child process: this one may have initialised the GPU already, and bench.py --gpus N starts its ranks before it touches one
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points", "100000", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
                        "--side-anchors", "0", "--scenes-in-flight", "0", "--skip-stages", "--skip-sizes", "--skip-v0"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL)")
def test_bench_on_two_ranks_over_rccl():
    out = _run_bench([])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["roundtrip_bit_identical"] is True
    ranks = out["ranks"]
    assert len(ranks) == 2
    assert ranks[0]["coded_nodes"] != ranks[1]["coded_nodes"], "both ranks coded the same scene"
    assert all(r["bytes"] > 0 and r["enc_ms"] > 0 and r["dec_ms"] > 0 for r in ranks)
    assert out["value"] > 0


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL)")
def test_batched_scenes_on_two_ranks_over_rccl():
    """BASELINE configs[3]: a batch of scenes per GPU, each rank's scenes through one chain of launches."""
    out = _run_bench(["--scenes-per-gpu", "4"])
    assert out["n_gpus"] == 2 and out["config"]["scenes_per_gpu"] == 4 and out["config"]["scenes_share_launches"] is True
    assert len(out["ranks"]) == 2 and out["ranks"][0]["coded_nodes"] != out["ranks"][1]["coded_nodes"]


def test_batched_scenes_per_gpu_on_one_rank():
    """The same entry point at N = 1 (runs on every GPU box): --scenes-per-gpu 4 codes the four scenes through one chain of launches."""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", "50000", "--scenes-per-gpu", "4", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
                        "--side-anchors", "0", "--scenes-in-flight", "0", "--skip-stages", "--skip-sizes", "--skip-v0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["scenes_share_launches"] is True and out["config"]["scenes_per_gpu"] == 4 and out["roundtrip_bit_identical"] is True
    assert out["ranks"][0]["coded_nodes"] > 4 * 100_000
