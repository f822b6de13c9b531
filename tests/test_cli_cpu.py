"""Host-side logic of the codec CLIs (reference: compress_ue_4stage_conv.py:38-62,90-95,245-279; kit/io.py:12-34)."""
import os
import struct

import numpy as np

from gauspcc_amd.cli import compress, decompress, io


def test_flag_names_and_defaults_match_reference():
    a = compress.build_parser().parse_args([])
    assert (a.input_glob, a.output_folder, a.is_data_pre_quantized, a.posQ, a.channels, a.kernel_size, a.num_samples, a.resultdir, a.prefix) == \
        ("./data/kittidet_examples/*.ply", "./data/kittidet_compressed/", False, 16, 32, 3, -1, "./results", "ue_4stage_conv")
    assert a.ckpt == "./model/KITTIDetection/ckpt_ue_4stage_conv.pt"
    d = decompress.build_parser().parse_args([])
    assert (d.channels, d.kernel_size, d.is_data_pre_quantized) == (32, 3, False)


def test_readers(tmp_path):
    rng = np.random.RandomState(0)
    pts = rng.randint(-500, 500, size=(50, 3)).astype(np.float32) * 0.25
    # KITTI .bin: float32 x,y,z,intensity
    np.concatenate([pts, rng.rand(50, 1).astype(np.float32)], 1).tofile(tmp_path / "a.bin")
    assert np.array_equal(io.read_points(str(tmp_path / "a.bin")), pts)
    # ASCII PLY (the layout kit/io.py:36-49 writes); header lines are skipped because they do not parse as numbers
    from gauspcc_amd.pcc_utils import save_ply_ascii_geo
    save_ply_ascii_geo(pts, str(tmp_path / "b.ply"))
    assert np.array_equal(io.read_points(str(tmp_path / "b.ply")).astype(np.float32), pts)
    # binary little-endian PLY with an extra property
    with open(tmp_path / "c.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nend_header\n")
        for p in pts:
            f.write(struct.pack("<fffB", *p, 7))
    assert np.array_equal(io.read_points(str(tmp_path / "c.ply")).astype(np.float32), pts)
    np.save(tmp_path / "d.npy", pts)
    assert np.array_equal(io.read_points(str(tmp_path / "d.npy")), pts)
    got = io.read_point_clouds([str(tmp_path / n) for n in ("a.bin", "d.npy")])
    assert len(got) == 2 and np.array_equal(got[0], got[1])


def test_input_listing(tmp_path):
    (tmp_path / "sub").mkdir()
    for n in ("sub/2.ply", "1.ply", "3.txt", "4.npy"):
        (tmp_path / n).write_text("x")
    got = [os.path.relpath(f, tmp_path) for f in compress.list_inputs(str(tmp_path))]
    assert got == ["1.ply", "4.npy", "sub/2.ply"]                 # sorted, filtered by extension (:56-58)
    assert len(compress.list_inputs(str(tmp_path), 2)) == 2       # first N, not random (:60-62)
    assert [os.path.basename(f) for f in compress.list_inputs(str(tmp_path / "*.ply"))] == ["1.ply"]
    # (the quantisation in front of the codec runs on the device: tests/test_gpu_parity.py::test_voxelise_*)


def test_results_csv(tmp_path):
    import pandas as pd
    rows = [{"filedir": "a.ply", "bpp": 2.0, "enc_time": 0.5, "file_size_bits": 200, "num_points": 100},
            {"filedir": "b.ply", "bpp": 4.0, "enc_time": 1.5, "file_size_bits": 800, "num_points": 200}]
    compress.write_results_csv(rows, tmp_path / "r.csv", with_avg=True)
    df = pd.read_csv(tmp_path / "r.csv")
    assert list(df.columns) == ["filedir", "bpp", "enc_time", "file_size_bits", "num_points"]
    assert df["filedir"].tolist() == ["a.ply", "b.ply", "avg"]
    assert df.iloc[2][["bpp", "enc_time", "file_size_bits", "num_points"]].tolist() == [3.0, 1.0, 500.0, 150.0]
