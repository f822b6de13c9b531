"""gauspcc_amd.torchac, the drop-in for the torchac 0.9.3 interface TC-GS / CAT-3DGS code their attributes with
(TC-GS/utils/encodings.py:84-176; HAC/utils/pcc_utils.py:174-177): ONE range-coder stream per call, coded by libgauspcc's
host-side twin of the device lane loop (csrc/hostcoder.hip).  Bytes == the oracle's restatement of the reference loop
(arithmetic_kernel.cu:94-163, 237-356); no GPU is involved when the tensors live on the CPU, as torchac's do."""
import time

import numpy as np
import pytest
import torch


def _table(n, lp, seed, sharp=1.0):
    rng = np.random.RandomState(seed)
    p = rng.dirichlet(np.ones(lp - 1) * sharp, size=n).astype(np.float32)
    cdf = np.concatenate([np.zeros((n, 1), np.float32), np.cumsum(p, 1)], 1).clip(0, 1).astype(np.float32)
    u = rng.rand(n, 1)
    sym = (u > np.cumsum(p, 1)).sum(1).clip(0, lp - 2).astype(np.int16)
    return cdf, sym


@pytest.mark.parametrize("lp,n,sharp", [(2, 1, 1.0), (3, 17, 1.0), (3, 50_000, 0.05), (5, 9_999, 0.3), (17, 20_000, 0.2), (65, 3_000, 1.0), (257, 777, 0.5)])
def test_bytes_equal_the_reference_loop(orc, lp, n, sharp):
    from gauspcc_amd import torchac

    cdf, sym = _table(n, lp, lp * 7 + n % 13, sharp)
    cdf_i = orc.cdf_to_int16(cdf)                                   # pinned against the reference's _convert_to_int_and_normalize (golden cdf_int.npz)
    ref = orc.rc_encode(cdf_i.view(np.uint16), sym.astype(np.uint8)) if lp <= 257 else None
    b1 = torchac.encode_float_cdf(torch.tensor(cdf), torch.tensor(sym), check_input_bounds=True)
    b2 = torchac.encode_int16_normalized_cdf(torch.tensor(cdf_i), torch.tensor(sym))
    assert b1 == ref and b2 == ref
    d1 = torchac.decode_float_cdf(torch.tensor(cdf), b1)
    d2 = torchac.decode_int16_normalized_cdf(torch.tensor(cdf_i), b2)
    assert d1.dtype == torch.int16 and np.array_equal(d1.numpy(), sym) and np.array_equal(d2.numpy(), sym)
    assert np.array_equal(orc.rc_decode(cdf_i.view(np.uint16), b1).astype(np.int16), sym)   # and the reference loop reads it back


def test_extreme_rows_long_carry_runs_and_shapes(orc):
    """Near-deterministic rows alternate between the two ends of the interval (long pending runs), rows like [0, 65535, 65536]
    renormalise to a span of 2^32; leading dimensions are flattened as torchac does."""
    from gauspcc_amd import torchac

    n = 6000
    cdf = np.zeros((n, 3), np.float32)
    cdf[:, 1] = np.where(np.arange(n) % 2 == 0, 1e-5, 1 - 1e-5)
    cdf[:, 2] = 1
    cdf_i = orc.cdf_to_int16(cdf)
    for sym in (np.zeros(n, np.int16), np.ones(n, np.int16), (np.arange(n) % 2).astype(np.int16)):
        b = torchac.encode_float_cdf(torch.tensor(cdf), torch.tensor(sym))
        assert b == orc.rc_encode(cdf_i.view(np.uint16), sym.astype(np.uint8))
        assert np.array_equal(torchac.decode_float_cdf(torch.tensor(cdf), b).numpy(), sym)
    cdf, sym = _table(6000, 9, 5)
    b3 = torchac.encode_float_cdf(torch.tensor(cdf).view(60, 100, 9), torch.tensor(sym).view(60, 100))
    assert b3 == torchac.encode_float_cdf(torch.tensor(cdf), torch.tensor(sym))
    assert torchac.decode_float_cdf(torch.tensor(cdf).view(60, 100, 9), b3).shape == (60, 100)
    assert torchac.encode_float_cdf(torch.zeros(0, 5), torch.zeros(0, dtype=torch.int16)) == b""
    with pytest.raises(ValueError):
        torchac.encode_float_cdf(torch.tensor(cdf), torch.tensor(sym.astype(np.int32)))
    with pytest.raises(Exception):
        torchac.encode_int16_normalized_cdf(torch.tensor(orc.cdf_to_int16(cdf)), torch.tensor(np.full(6000, 8, np.int16)))   # symbol == Lp - 1


def test_float_rows_on_the_fly_equal_torch_rounding(orc):
    """A float32 table on the host is integerised inside the coder (gsac_host_*_f32); every other table goes through torch ops
    (_to_int_rows).  Same rows, ties included: cdf values placed exactly on k + 0.5 of the scaled grid round half to even in both."""
    from gauspcc_amd import torchac

    lp, n = 9, 4000
    scale = float(2 ** 16 - (lp - 1))
    rng = np.random.RandomState(5)
    k = np.sort(rng.randint(0, 65000, size=(n, lp - 2)), axis=1).astype(np.float64)
    inner = ((k + 0.5) / scale).astype(np.float32)                 # as close to a tie as fp32 gets; many are exact ties after the fp32 multiply
    cdf = np.concatenate([np.zeros((n, 1), np.float32), inner, np.ones((n, 1), np.float32)], 1)
    sym = rng.randint(0, lp - 1, size=n).astype(np.int16)
    t32 = torch.tensor(cdf)
    rows_torch = torchac._to_int_rows(t32).numpy()
    assert np.array_equal(rows_torch, orc.cdf_to_int16(cdf))
    b_fly = torchac.encode_float_cdf(t32, torch.tensor(sym))       # float32 on the host: on the fly
    b_rows = torchac.encode_int16_normalized_cdf(torch.tensor(rows_torch), torch.tensor(sym))
    assert b_fly == b_rows
    # a float64 table is multiplied and rounded in float64, as torchac does with the tensor it is given (its _convert_to_int_and_normalize has no
    # cast): the fp32 ties above are not ties in fp64, so the rows -- and the bytes -- are those of the float64 arithmetic
    t64 = t32.double()
    rows64 = ((t64 * scale).round().to(torch.int32) + torch.arange(lp, dtype=torch.int32)).to(torch.int16)
    assert torch.equal(torchac._to_int_rows(t64), rows64)
    b_f64 = torchac.encode_float_cdf(t64, torch.tensor(sym))
    assert b_f64 == torchac.encode_int16_normalized_cdf(rows64, torch.tensor(sym))
    assert torch.equal(torchac.decode_float_cdf(t64, b_f64), torch.tensor(sym))
    assert np.array_equal(torchac.decode_float_cdf(t32, b_fly).numpy(), torchac.decode_int16_normalized_cdf(torch.tensor(rows_torch), b_fly).numpy())


def test_faster_than_one_gpu_lane():
    """Round 3's shim ran the single stream on one GPU lane: 4.8 / 2.2 Msymbols/s.  The host loop must beat that on any core
    (the oracle's bit-by-bit restatement does ~15 / 10 on the build box)."""
    from gauspcc_amd import torchac

    cdf, sym = _table(400_000, 17, 3, 0.3)
    ci = torch.tensor(cdf)
    rows = torchac._to_int_rows(ci)
    s = torch.tensor(sym)
    t0 = time.perf_counter()
    b = torchac.encode_int16_normalized_cdf(rows, s)
    t1 = time.perf_counter()
    d = torchac.decode_int16_normalized_cdf(rows, b)
    t2 = time.perf_counter()
    assert torch.equal(d, s)
    assert 0.4 / (t1 - t0) > 5.0 and 0.4 / (t2 - t1) > 4.0, (0.4 / (t1 - t0), 0.4 / (t2 - t1))
