"""ADVICE (round 5, medium): the chunk byte-count table of a `.b` slice file comes off disk and the library reads one count per chunk
of every stream it decodes; a file whose header states a shorter table made it read past the numpy buffer on the host.  The table's
length is now checked against the symbol count at the two layers every `.b` decoder goes through (gauspcc_amd/arithmetic.py:
_check_cnt; gauspcc_amd/encodings_cuda.py: _read_slice_files) -- both before any pointer crosses the C ABI, so this runs without a GPU.
File layout: HAC/utils/encodings_cuda.py:366-376 (f32 min | f32 max | i32 table bytes | table | payload)."""
import numpy as np
import pytest


def _slice_file(path, n_cnt, lc=None, payload=40):
    cnt = np.full(n_cnt, payload // max(n_cnt, 1), dtype=np.int32)
    head = np.float32(-3).tobytes() + np.float32(3).tobytes() + np.int32(4 * n_cnt if lc is None else lc).tobytes()
    with open(path, "wb") as f:
        f.write(head + cnt.tobytes() + bytes(payload))
    return str(path)


def test_check_cnt_rejects_short_long_and_negative_tables():
    from gauspcc_amd.arithmetic import _check_cnt

    _check_cnt(np.zeros(3, np.int32), [25_000], 10_000, "t")
    _check_cnt(np.zeros(4, np.int32), [25_000, 1], 10_000, "t")
    _check_cnt(np.zeros(0, np.int32), [0], 10_000, "t")
    for bad in (np.zeros(2, np.int32), np.zeros(4, np.int32)):
        with pytest.raises(ValueError, match="chunk table"):
            _check_cnt(bad, [25_000], 10_000, "t")
    with pytest.raises(ValueError, match="negative"):
        _check_cnt(np.array([5, -1, 2], np.int32), [25_000], 10_000, "t")
    with pytest.raises(ValueError):
        _check_cnt(np.zeros(1, np.int32), [10], 0, "t")


def test_read_slice_files_checks_every_table_against_its_slice(tmp_path):
    from gauspcc_amd.encodings_cuda import _read_slice_files

    good = [_slice_file(tmp_path / "a_0.b", 1), _slice_file(tmp_path / "b_0.b", 2)]
    mins, maxs, cnts, datas = _read_slice_files(good, [3000, 12_000])
    assert [c.size for c in cnts] == [1, 2] and mins[0] == -3 and maxs[1] == 3 and datas[0].size == 40
    with pytest.raises(RuntimeError, match="chunk table"):                 # header states a shorter table than the slice needs
        _read_slice_files(good, [3000, 25_000])
    with pytest.raises(RuntimeError, match="chunk table"):                 # ... or a longer one
        _read_slice_files(good, [3000, 3000])
    with pytest.raises(RuntimeError, match="bad chunk table"):             # not a whole number of int32
        _read_slice_files([_slice_file(tmp_path / "c_0.b", 1, lc=6)], [3000])
    with pytest.raises(RuntimeError, match="bad chunk table"):             # negative / past the end of the file
        _read_slice_files([_slice_file(tmp_path / "d_0.b", 1, lc=-4)], [3000])
    with pytest.raises(RuntimeError, match="bad chunk table"):
        _read_slice_files([_slice_file(tmp_path / "e_0.b", 1, lc=4000)], [3000])
    short = tmp_path / "f_0.b"
    short.write_bytes(b"\x00" * 7)
    with pytest.raises(RuntimeError, match="truncated"):
        _read_slice_files([str(short)], [3000])
