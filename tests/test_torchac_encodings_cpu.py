"""gauspcc_amd.torchac_encodings: the torchac-based attribute coders of TC-GS / CAT-3DGS (TC-GS/utils/encodings.py:84-183,
CAT-3DGS/utils/encodings.py:39-175).  Expected bytes come from the reference's own statement of the
table -- `Normal(mean, scale).cdf((samples - 0.5) * Q)` over samples = min .. max + 1, written out here in plain torch --
pushed through the oracle's restatement of torchac's integerisation and coder loop.  CPU tensors: no GPU involved."""
import numpy as np
import pytest
import torch


def _scene(n, seed):
    g = torch.Generator().manual_seed(seed)
    mean = torch.randn(n, generator=g) * 0.8
    scale = torch.rand(n, generator=g) * 0.9 + 0.05
    Q = torch.rand(n, generator=g) * 0.1 + 0.05
    x = mean + scale * torch.randn(n, generator=g)
    x = x.clamp(-3.0, 3.0)
    return x, mean, scale, Q


def _reference_table(mean, scale, Q, lo, hi):
    samples = torch.tensor(range(int(lo), int(hi) + 1 + 1)).to(torch.float)             # encodings.py:92-93
    samples = samples.unsqueeze(0).repeat(mean.shape[0], 1)
    m = mean.unsqueeze(-1).repeat(1, samples.shape[-1])
    s = scale.unsqueeze(-1).repeat(1, samples.shape[-1])
    return torch.distributions.normal.Normal(m, s).cdf((samples - 0.5) * Q.unsqueeze(-1))   # :97-98


@pytest.mark.parametrize("n,seed", [(1, 1), (257, 2), (20_000, 3)])
def test_gaussian_coder_bytes_and_roundtrip(orc, tmp_path, n, seed, monkeypatch):
    from gauspcc_amd import torchac_encodings as te

    x, mean, scale, Q = _scene(n, seed)
    f = str(tmp_path / "a.b")
    bits, lo, hi = te.encoder_gaussian(x, mean, scale, Q, file_name=f)
    data = open(f, "rb").read()
    assert bits == 8 * len(data)
    xi = torch.round(x / Q)
    assert lo == xi.min() and hi == xi.max()
    lower = _reference_table(mean, scale, Q, lo.item(), hi.item())
    lp = lower.shape[1]
    assert lp <= 257
    rows = orc.cdf_to_int16(lower.numpy())
    assert data == orc.rc_encode(rows.view(np.uint16), (xi - lo).numpy().astype(np.uint8))
    dec = te.decoder_gaussian(mean, scale, Q, file_name=f, min_value=lo, max_value=hi)
    assert dec.dtype == torch.float32 and torch.equal(dec, xi * Q)
    # the table in slabs of a few rows: the same file
    monkeypatch.setattr(te, "_SLAB_ENTRIES", 5 * lp)
    f2 = str(tmp_path / "b.b")
    te.encoder_gaussian(x, mean, scale, Q, file_name=f2)
    assert open(f2, "rb").read() == data
    assert torch.equal(te.decoder_gaussian(mean, scale, Q, file_name=f2, min_value=lo.item(), max_value=hi.item()), xi * Q)


def test_scalar_q_and_binary_coder(orc, tmp_path):
    from gauspcc_amd import torchac_encodings as te

    x, mean, scale, _ = _scene(3000, 7)
    f = str(tmp_path / "q.b")
    bits, lo, hi = te.encoder_gaussian(x, mean, scale, 0.125, file_name=f)              # Q as a number: encodings.py:86-87
    dec = te.decoder_gaussian(mean, scale, 0.125, file_name=f, min_value=lo, max_value=hi)
    assert torch.equal(dec, torch.round(x / 0.125) * 0.125)
    # binary masks, encodings.py:149-183
    g = torch.Generator().manual_seed(11)
    p = torch.rand(50_000, generator=g).clamp(0.01, 0.99)
    xb = (torch.rand(50_000, generator=g) < p).to(torch.float32) * 2 - 1
    fb = str(tmp_path / "m.b")
    bits = te.encoder(xb, p, fb)
    data = open(fb, "rb").read()
    assert bits == 8 * len(data)
    p_u = 1 - p.unsqueeze(-1)
    cdf = torch.cat([torch.zeros_like(p_u), p_u, torch.ones_like(p_u)], dim=-1)
    rows = orc.cdf_to_int16(cdf.numpy())
    assert data == orc.rc_encode(rows.view(np.uint16), torch.floor((xb + 1) / 2).numpy().astype(np.uint8))
    assert torch.equal(te.decoder(p, fb), xb)
    # the stream costs what the model says these symbols cost (16-bit counts: a fraction of a per cent on top)
    ideal = float(-torch.log2(torch.where(xb > 0, p, 1 - p)).sum())
    assert ideal <= bits <= ideal * 1.01 + 64


def test_degenerate_ranges_and_empty_masks(orc, tmp_path):
    """All symbols equal (the table has two entries + the closing one), a single element, and a mask of one value only."""
    from gauspcc_amd import torchac_encodings as te

    n = 500
    mean = torch.zeros(n); scale = torch.full((n,), 0.3); Q = torch.full((n,), 0.1)
    x = torch.full((n,), 0.7)
    f = str(tmp_path / "c.b")
    bits, lo, hi = te.encoder_gaussian(x, mean, scale, Q, file_name=f)
    assert lo == hi == 7.0
    assert torch.equal(te.decoder_gaussian(mean, scale, Q, file_name=f, min_value=lo, max_value=hi), torch.round(x / Q) * Q)
    f1 = str(tmp_path / "one.b")
    bits, lo, hi = te.encoder_gaussian(x[:1], mean[:1], scale[:1], 0.1, file_name=f1)
    assert torch.equal(te.decoder_gaussian(mean[:1], scale[:1], 0.1, file_name=f1, min_value=lo, max_value=hi), torch.round(x[:1] / 0.1) * 0.1)
    p = torch.full((300,), 0.9)
    xb = torch.ones(300)
    fb = str(tmp_path / "m.b")
    bits = te.encoder(xb, p, fb)
    assert bits <= 8 * 16 and torch.equal(te.decoder(p, fb), xb)       # 300 x 0.15 bits


@pytest.mark.parametrize("n", [3, 25_001])
def test_ten_way_fan_out_writes_the_reference_chunk_files(orc, tmp_path, n, monkeypatch):
    """use_multiprocessor = True (encodings.py:14, :36-82, :114, :136): rows [m c, (m + 1) c), c = ceil(n / 10), of the table and their symbols are
    one torchac stream each in `<name>_<m>.b` -- every file equals the oracle's coder on that chunk of the reference's table, the decoder
    concatenates the chunks, and the total bit length is the sum.  Ten native threads; a chunk past the end is an empty file."""
    import os

    from gauspcc_amd import torchac_encodings as te

    x, mean, scale, Q = _scene(n, 11)
    monkeypatch.setattr(te, "use_multiprocessor", True)
    f = str(tmp_path / "feat.b")
    bits, lo, hi = te.encoder_gaussian(x, mean, scale, Q, file_name=f)
    xi = torch.round(x / Q)
    lower = _reference_table(mean, scale, Q, lo.item(), hi.item())
    rows = orc.cdf_to_int16(lower.numpy()).view(np.uint16)
    sym = (xi - lo).numpy().astype(np.uint8)
    c = -(-n // 10)
    total = 0
    for m in range(10):
        data = open(str(tmp_path / f"feat_{m}.b"), "rb").read()
        a, b = m * c, min(n, (m + 1) * c)
        assert data == (orc.rc_encode(rows[a:b], sym[a:b]) if b > a else b""), m
        total += 8 * len(data)
    assert bits == total and not os.path.exists(f)
    dec = te.decoder_gaussian(mean, scale, Q, file_name=f, min_value=lo, max_value=hi)
    assert dec.dtype == torch.float32 and torch.equal(dec, xi * Q)
    # the stand-alone entry points on a float table, as the reference calls them (:114, :136)
    f2 = str(tmp_path / "g.b")
    assert te.multiprocess_encoder(lower, (xi - lo).to(torch.int16), f2) == total
    assert torch.equal(te.multiprocess_deoder(lower, f2, chunk_num=10), (xi - lo).to(torch.float32))
    for m in range(10):
        assert open(str(tmp_path / f"g_{m}.b"), "rb").read() == open(str(tmp_path / f"feat_{m}.b"), "rb").read()
