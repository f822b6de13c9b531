"""GPU tests of the batch axis (gpcc_encode_batch / gpcc_decode_batch, csrc/forest.hpp): K scenes through ONE chain of
launches.  The bar: every scene's container is byte-identical to its solo encode and to the
oracle's, and a batch decodes to the solo decodes' points in the solo order.  Reference: the batch column of the codec's
coordinates (HAC/utils/pcc_utils.py:73, kit/op.py:17-30) and the CLI's file loop (compress_ue_4stage_conv.py:72-75)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gh():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X: the HIP path has no fallback")
    from tests import gpu_helpers

    return gpu_helpers


def _dev_model(k):
    from gauspcc_amd import runtime
    from gauspcc_amd.synth import synthetic_state_dict

    return runtime.Model(synthetic_state_dict(32, k), 32, k, 0)


@pytest.fixture(scope="module")
def dev_model_k5(gh):
    return _dev_model(5)


@pytest.fixture(scope="module")
def dev_model_k3(gh):
    return _dev_model(3)


def _cloud(n, seed=1234, negative=False, extent_log2=16):
    from gauspcc_amd.synth import synthetic_cloud

    return synthetic_cloud(n, seed=seed, negative=negative, extent_log2=extent_log2)


def _enc_batch(gh, model, clouds, chunk_log2=11, posq=None, version=None):
    import torch

    from gauspcc_amd.pcc_utils import _encode_batch

    xs = [torch.tensor(np.ascontiguousarray(c, dtype=np.int32), device=gh.dev()) for c in clouds]
    return _encode_batch(xs, model, chunk_log2, posq or [1] * len(xs), version)


def _dec_batch(gh, model, datas):
    from gauspcc_amd.pcc_utils import _decode_batch

    outs, pqs, sts, batched = _decode_batch(datas, model, gh.dev())
    return [o.cpu().numpy() for o in outs], pqs, sts, batched


def _check_batch(gh, model, clouds, chunk_log2=11, orc=None, omodel=None, expect_batched=True, version=None, expect_dec_batched=None):
    blobs, stats, batched = _enc_batch(gh, model, clouds, chunk_log2, version=version)
    assert batched == expect_batched
    solo = []
    for i, c in enumerate(clouds):
        data, st = gh.encode(model, c, chunk_log2, version=version)
        assert len(blobs[i]) == len(data), (i, len(blobs[i]), len(data))
        assert blobs[i] == data, f"scene {i}: first differing byte at {next(j for j, (a, b) in enumerate(zip(blobs[i], data)) if a != b)}"
        assert stats[i].num_points == len(c) and stats[i].num_bytes == len(data) and stats[i].num_levels == st.num_levels
        assert list(stats[i].level_nodes[: st.num_levels]) == list(st.level_nodes[: st.num_levels])
        solo.append(data)
        if orc is not None:
            assert data == orc.encode(omodel, c, chunk_log2=chunk_log2)
    outs, pqs, sts, dbatched = _dec_batch(gh, model, blobs)
    assert dbatched == ((expect_batched and chunk_log2 != 0) if expect_dec_batched is None else expect_dec_batched)
    for i, c in enumerate(clouds):
        dec, _, _ = gh.decode(model, solo[i])
        assert outs[i].shape == dec.shape
        assert np.array_equal(outs[i], dec), f"scene {i}: decoded points differ from the solo decode"
        assert float(pqs[i]) == 1.0 and sts[i].num_points == len(c)
    return blobs


def test_batch_of_one_is_the_solo_encode(gh, orc, dev_model_k5, synth_model_k5):
    _check_batch(gh, dev_model_k5, [_cloud(10_000, seed=5)], orc=orc, omodel=synth_model_k5)


@pytest.mark.parametrize("k", [5, 3])
def test_small_batch_bytes_identical_to_solo_and_oracle(gh, orc, k, dev_model_k5, dev_model_k3, synth_model_k5, synth_model_k3):
    dm, om = (dev_model_k5, synth_model_k5) if k == 5 else (dev_model_k3, synth_model_k3)
    clouds = [_cloud(10_000, seed=21), _cloud(4_000, seed=22, negative=True), _cloud(25_000, seed=23, extent_log2=14)]
    _check_batch(gh, dm, clouds, orc=orc, omodel=om)


def test_eight_scenes_of_100k(gh, dev_model_k5):
    """8 x 100 k mixed seeds."""
    _check_batch(gh, dev_model_k5, [_cloud(100_000, seed=100 + i) for i in range(8)])


def test_five_scenes_of_mixed_depth(gh, orc, dev_model_k5, synth_model_k5):
    """7 / 3 k / 40 k / 250 k / 1 M points: scenes that end at different depths of the merged tree, one with no coded level at all."""
    clouds = [_cloud(7, seed=31), _cloud(3_000, seed=32, extent_log2=12), _cloud(40_000, seed=33), _cloud(250_000, seed=34), _cloud(1_000_000, seed=1234)]
    blobs = _check_batch(gh, dev_model_k5, clouds)
    for i in (0, 1, 2):
        assert blobs[i] == orc.encode(synth_model_k5, clouds[i], chunk_log2=11)


def test_two_scenes_of_one_million(gh, dev_model_k5):
    _check_batch(gh, dev_model_k5, [_cloud(1_000_000, seed=1234), _cloud(1_000_000, seed=77)])


def test_thirty_two_small_scenes(gh, dev_model_k5):
    _check_batch(gh, dev_model_k5, [_cloud(10_000, seed=300 + i, negative=bool(i & 1)) for i in range(32)])


def test_far_and_negative_scenes_share_a_tree(gh, dev_model_k3):
    """Scenes anywhere in int32 (each keeps its own frame; only the extents share the 21-bit budget)."""
    a = _cloud(6_000, seed=41) + np.array([1_900_000_000, -7, 12_345], dtype=np.int32)
    b = _cloud(9_000, seed=42, negative=True) - np.array([1_000_000_000, 2_000_000_000, 3], dtype=np.int32)
    c = _cloud(5_000, seed=43, extent_log2=10)
    _check_batch(gh, dev_model_k3, [a, b, c])


def test_version_three_batch(gh, dev_model_k5):
    _check_batch(gh, dev_model_k5, [_cloud(12_000, seed=51), _cloud(30_000, seed=52)], chunk_log2=9, version=3)
    gh.set_version(4)


def test_reference_layout_falls_back_to_solo(gh, dev_model_k3):
    """chunk_log2 = 0 (the reference container) has no level table: the scenes are coded one by one, same bytes."""
    _check_batch(gh, dev_model_k3, [_cloud(3_000, seed=61), _cloud(2_000, seed=62)], chunk_log2=0, expect_batched=False)


def test_wide_scene_falls_back_to_solo(gh, dev_model_k3):
    """A scene whose extent reaches 2^20 cannot share a frame: the batch entry point codes the scenes one by one."""
    wide = _cloud(4_000, seed=71, extent_log2=20, negative=True)
    # (the decoder picks its own frame from the base levels in the headers, and those DO stack: it batches)
    _check_batch(gh, dev_model_k3, [_cloud(3_000, seed=72), wide], expect_batched=False, expect_dec_batched=True)


def test_duplicate_point_names_its_scene(gh, dev_model_k3):
    from gauspcc_amd import _lib

    a = _cloud(2_000, seed=81)
    b = _cloud(2_000, seed=82)
    b[5] = b[9]
    with pytest.raises(_lib.GpccError, match="scene 1"):
        _enc_batch(gh, dev_model_k3, [a, b])
    # the context still works
    _check_batch(gh, dev_model_k3, [a, _cloud(2_000, seed=82)])


def test_corrupt_scene_in_a_batch_is_an_error_not_a_crash(gh, dev_model_k3):
    from gauspcc_amd import _lib

    clouds = [_cloud(20_000, seed=91), _cloud(15_000, seed=92)]
    blobs, _, _ = _enc_batch(gh, dev_model_k3, clouds)
    rng = np.random.default_rng(7)
    for trial in range(12):
        bad = bytearray(blobs[1])
        if trial % 3 == 0:
            i = int(rng.integers(8, 8 + 4 * bad[6]))          # a level size of the header
            bad[i] ^= 1 << int(rng.integers(0, 8))
        elif trial % 3 == 1:
            for _ in range(40):                              # payload bytes
                bad[int(rng.integers(len(bad) // 2, len(bad)))] ^= int(rng.integers(1, 256))
        else:
            del bad[int(rng.integers(len(bad) // 2, len(bad))):]
        try:
            outs, _, _, _ = _dec_batch(gh, dev_model_k3, [blobs[0], bytes(bad)])
        except _lib.GpccError:
            continue
        assert outs[0].shape[1] == 3   # garbage in, some cloud out: never a fault
    outs, _, _, _ = _dec_batch(gh, dev_model_k3, blobs)
    assert np.array_equal(outs[0], gh.decode(dev_model_k3, blobs[0])[0])


def test_plugin_api_batch_roundtrip(gh, tmp_path):
    import torch

    from gauspcc_amd import pcc_utils

    ckpt = "synthetic"
    clouds = [_cloud(5_000, seed=3), _cloud(8_000, seed=4)]
    paths = [str(tmp_path / f"s{i}.bin") for i in range(2)]
    res = pcc_utils.compress_point_clouds([torch.tensor(c) for c in clouds], ckpt, paths, kernel_size=3)
    assert [r['num_points'] for r in res] == [5_000, 8_000]
    solo = pcc_utils.compress_point_cloud(torch.tensor(clouds[1]), ckpt, str(tmp_path / "solo.bin"), kernel_size=3)
    assert open(paths[1], 'rb').read() == open(solo['output_path'], 'rb').read()
    dec = pcc_utils.decompress_point_clouds(paths, ckpt, kernel_size=3)
    for c, d in zip(clouds, dec):
        got = d['point_cloud'].cpu().numpy().astype(np.int64)
        assert np.array_equal(got[np.lexsort((got[:, 0], got[:, 1], got[:, 2]))], c[np.lexsort((c[:, 0], c[:, 1], c[:, 2]))])
    one = pcc_utils.decompress_point_cloud(paths[0], ckpt, kernel_size=3)
    assert torch.equal(one['point_cloud'], dec[0]['point_cloud'])


def test_cli_batch_gives_the_same_files(gh, tmp_path):
    """--batch 3 (extension: three files through one chain of launches) writes byte-identical .bin files, the same CSV rows in the
    same order, and PLYs with the same points as the one-at-a-time run (reference file loop: compress_ue_4stage_conv.py:72-75)."""
    import pandas as pd
    from gauspcc_amd.cli import compress, decompress, io
    from gauspcc_amd.pcc_utils import save_ply_ascii_geo

    src = tmp_path / "src"
    src.mkdir()
    for i, n in enumerate((20000, 3000, 12000, 7000, 16000)):
        save_ply_ascii_geo(_cloud(n, seed=70 + i).astype(np.float32), str(src / f"c{i}.ply"))
    common = ["--channels", "32", "--kernel_size", "3", "--ckpt", "synthetic:3"]
    tabs, recs = [], []
    for batch in (1, 3):
        out, rec, res = (tmp_path / f"{n}{batch}" for n in ("bin", "rec", "res"))
        assert compress.main(["--input_glob", str(src), "--output_folder", str(out), "--is_data_pre_quantized", "1", "--posQ", "1",
                              "--resultdir", str(res), "--prefix", "t", "--batch", str(batch)] + common) == 0
        assert decompress.main(["--input_glob", str(out / "*.bin"), "--output_folder", str(rec), "--is_data_pre_quantized", "1", "--batch", str(batch)] + common) == 0
        tabs.append(pd.read_csv(res / "t_data5.csv"))
        recs.append((out, rec))
    assert tabs[0]["filedir"].tolist() == tabs[1]["filedir"].tolist() == [f"c{i}.ply" for i in range(5)] + ["avg"]
    assert tabs[0]["file_size_bits"].tolist() == tabs[1]["file_size_bits"].tolist()
    for i in range(5):
        a, b = (open(o / f"c{i}.ply.bin", "rb").read() for o, _ in recs)
        assert a == b
        pa, pb = (io.read_points(str(r / f"c{i}.ply.bin.ply")) for _, r in recs)
        assert np.array_equal(pa, pb)


def test_identical_and_tiny_scenes_share_a_tree(gh, orc, dev_model_k3, synth_model_k3):
    """Edge cases of the merged tree: the SAME cloud three times (the scenes' own frames coincide; only the z slabs keep them apart), a single
    point, seven points (a base level and nothing coded), a cloud of 64 points (one coded level) -- all in one batch, every container == the oracle's."""
    same = _cloud(3_000, seed=401)
    clouds = [same, np.array([[5, -3, 11]], dtype=np.int32), same.copy(), _cloud(7, seed=402), _cloud(64, seed=403, extent_log2=6), same.copy(), _cloud(200, seed=404, extent_log2=17, negative=True)[:100]]
    blobs = _check_batch(gh, dev_model_k3, clouds)
    for b, c in zip(blobs, clouds):
        assert b == orc.encode(synth_model_k3, c, chunk_log2=11)
    assert blobs[0] == blobs[2] == blobs[5]


def test_more_scenes_than_a_tree_holds(gh, dev_model_k3):
    """257 scenes exceed FOREST_MAX_SCENES: the entry points code them one by one (batched flag 0), same bytes; 256 share a tree."""
    clouds = [_cloud(40 + (i % 7) * 30, seed=500 + i, extent_log2=8) for i in range(257)]
    blobs, _, batched = _enc_batch(gh, dev_model_k3, clouds)
    assert not batched
    b256, _, batched256 = _enc_batch(gh, dev_model_k3, clouds[:256])
    assert batched256 and b256 == blobs[:256]
    outs, _, _, _ = _dec_batch(gh, dev_model_k3, b256)
    for o, c in zip(outs[::17], clouds[:256:17]):
        assert np.array_equal(o[np.lexsort((o[:, 0], o[:, 1], o[:, 2]))], c[np.lexsort((c[:, 0], c[:, 1], c[:, 2]))])


def test_random_batch_shapes_against_the_solo_path():
    """tools/batch_soak.py for 25 s: random batches (1-40 scenes of 1-300 k points, extents 2^6-2^17, shifted clouds, k 3 / 5, chunk_log2 6-11) --
    every scene's batched bytes == its solo bytes, the batched decode == the solo decode == the input set (reference loop:
    HAC/utils/pcc_utils.py:73-131 once per scene)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "batch_soak.py"), "25", "7"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 bad" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert int(r.stdout.strip().splitlines()[-1].split()[2]) >= 20      # batches run
