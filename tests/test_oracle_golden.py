"""Pin the oracle to the reference: every golden vector captured by
tests/golden/make_golden.py (reference functions run in the build container)."""
import numpy as np
import pytest

MORTON_CASES = ["cube_small", "negative", "flat_x", "noncubic_int", "wide", "single"]


@pytest.mark.parametrize("name", MORTON_CASES)
def test_raster_order_matches_reference(orc, golden_dir, name):
    z = np.load(f"{golden_dir}/morton.npz")
    perm = orc.raster_order(z[f"{name}_in"])
    assert perm.dtype == np.int64
    assert np.array_equal(perm, z[f"{name}_perm"])


@pytest.mark.parametrize("i", [0, 1, 2])
def test_sort_cf_matches_reference(orc, golden_dir, i):
    z = np.load(f"{golden_dir}/sort_cf.npz")
    c, f = orc.sort_cf(z[f"c{i}_in"], z[f"f{i}_in"])
    assert np.array_equal(c, z[f"c{i}_out"])
    assert np.array_equal(f, z[f"f{i}_out"])
    assert np.array_equal(c, z[f"c{i}_sortC"])


@pytest.mark.parametrize("lp", [3, 5, 17])
def test_cdf_integerisation_matches_reference(orc, golden_dir, lp):
    z = np.load(f"{golden_dir}/cdf_int.npz")
    out = orc.cdf_to_int16(z[f"lp{lp}_in"])
    assert np.array_equal(out, z[f"lp{lp}_out"])


def test_pack_unpack_matches_reference(orc, golden_dir):
    z = np.load(f"{golden_dir}/pack.npz")
    streams = [z[f"s{i}"].tobytes() for i in range(4)]
    assert orc.pack_byte_stream_ls(streams) == z["packed"].tobytes()
    assert orc.unpack_byte_stream(z["packed"].tobytes()) == streams


def test_psnr_matches_reference(orc, golden_dir):
    z = np.load(f"{golden_dir}/image.npz")
    np.testing.assert_allclose(orc.psnr(z["a"], z["b"]), z["psnr"], rtol=1e-6)


def test_raster_order_equals_zyx_lexsort(orc):
    """F2 of SURVEY.md: the 'Morton' order is the (z, y, x) raster order -- which is
    also what sort_CF produces inside the codec."""
    rng = np.random.RandomState(3)
    p = np.unique(rng.randint(-50, 50, (3000, 3)), axis=0)
    p = p[rng.permutation(len(p))]
    perm = orc.raster_order(p.astype(np.float32))
    assert np.array_equal(perm, np.lexsort((p[:, 0], p[:, 1], p[:, 2])))
