"""gpcc_write_files: the native writer behind the per-slice `.b` files of the attribute loops (gauspcc_amd/encodings_cuda.py:
_write_files; HAC/scene/gaussian_model.py:1176-1213 writes one file per 3 000-anchor slice and attribute).  No GPU involved."""
import os

import pytest


def test_write_files_contents_empty_files_and_errors(tmp_path):
    from gauspcc_amd import encodings_cuda as ec
    from gauspcc_amd._lib import GpccError

    jobs = [(str(tmp_path / f"s_{i}_0.b"), bytes([(i * 31 + k) % 256 for k in range(i * 13 % 4000)])) for i in range(700)]
    jobs.append((str(tmp_path / "nul.b"), b"\x00\x00a\x00"))         # embedded NULs: the sizes count, not a terminator
    jobs.append((str(tmp_path / "empty.b"), b""))
    ec._write_files(jobs)
    for path, blob in jobs:
        with open(path, "rb") as f:
            assert f.read() == blob, path
    ec._write_files([])                                               # nothing to do
    with pytest.raises(GpccError) as e:
        ec._write_files([(str(tmp_path / "ok.b"), b"x"), (str(tmp_path / "no_such_dir" / "x.b"), b"abc")])
    assert "no_such_dir" in str(e.value)
    assert os.path.exists(tmp_path / "ok.b")                          # the other files of a failing call are still written
