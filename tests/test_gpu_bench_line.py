"""bench.py end to end on the GPU box, at a reduced point count: the ONE JSON line the driver parses carries the contract keys, the
`roofline` and `cpu_baseline` objects, and every untimed pass behind the timed region ran (sizes, batched, reference_layout, low_rate,
chunk_sweep, scenes_in_flight, side_paths) -- a pass that raises takes the whole line with it, and nothing else in the suite runs the
script as the driver does.  (Round 6: a function-local `import tempfile` in tools/bench_side_paths.py shadowed the module's and broke
exactly this path; no test saw it.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_prints_one_complete_json_line():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--points", "100000", "--steps", "3", "--warmup", "1", "--cpu-sample", "20000",
                        "--side-anchors", "20000"], capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]               # stdout carries the one JSON line only
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["unit"] == "Mpoints/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 100000 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 0.02 * d["value"]
    assert d["roundtrip_bit_identical"] is True
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "Mpoints/s" and "sample" in cb
    assert [s["points"] for s in d["sizes"]] == [10_000, 100_000, 1_000_000]
    assert all(b["bytes_identical_to_solo"] and b["decode_identical_to_solo"] and b["one_tree"] for b in d["batched"])
    ref = d["reference_layout"]
    assert ref["roundtrip_bit_identical"] and ref["bytes"] == d["bytes_v0"] and ref["dec_ms"] > ref["enc_ms"] > 0
    assert len(d["low_rate"]) == 2 and all(c["roundtrip_bit_identical"] and c["bytes_v0"] <= c["container_bytes"] for c in d["low_rate"])
    assert d["low_rate"][0]["bpp"] < 0.8 * d["bpp"] and d["low_rate"][1]["bits_per_coded_node"] < 6.0
    sw = d["chunk_sweep"]
    assert [c["chunk_log2"] for c in sw] == [9, 10, 11, 12, 13] and all(c["roundtrip_bit_identical"] for c in sw)
    assert all(a["bytes"] >= b["bytes"] for a, b in zip(sw, sw[1:]))              # smaller chunks cost bytes
    assert d["scenes_in_flight"]["scenes"] == 2 and d["scenes_in_flight"]["value"] > 0
    sp = d["side_paths"]
    assert sp["torchac_shim"]["roundtrip"] and sp["torchac_shim"]["fan_out_roundtrip"] and sp["gaussian_coder"]["fused_bytes_equal_table_bytes"]
    assert sp["rd_loop"]["psnr_decoded_vs_encoder_side_dB"] > 40
