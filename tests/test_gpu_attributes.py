"""GPU parity of the HAC attribute-side kernels against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    return torch


def test_calculate_cdf_matches_oracle(torch_cuda, orc):
    torch = torch_cuda
    from gauspcc_amd import arithmetic

    rng = np.random.RandomState(0)
    n = 5000
    mean = rng.randn(n).astype(np.float32) * 3
    scale = np.abs(rng.randn(n)).astype(np.float32) * 2 + 1e-3
    scale[:5] = 0.0                     # clamped to 1e-9
    q = (rng.rand(n).astype(np.float32) + 0.5)
    lower = arithmetic.calculate_cdf(torch.tensor(mean).cuda(), torch.tensor(scale).cuda(), torch.tensor(q).cuda(), -12, 15).cpu().numpy()
    ref = orc.gaussian_cdf(mean, scale, q, -12, 15)
    assert lower.shape == ref.shape == (n, 29)
    # erfc comes from two different math libraries (ocml vs glibc): a few ulp, far below one CDF quantum (1/65536)
    np.testing.assert_allclose(lower, ref, atol=2e-7, rtol=0)


@pytest.mark.parametrize("lp,chunk", [(3, 10000), (30, 10000), (30, 777), (100, 5000)])
def test_hac_coder_bytes_and_roundtrip(torch_cuda, orc, lp, chunk):
    torch = torch_cuda
    from gauspcc_amd import arithmetic

    rng = np.random.RandomState(lp + chunk)
    n = 23456
    logits = rng.randn(n, lp - 1).astype(np.float32) * 2
    p = np.exp(logits - logits.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    cdf = np.concatenate([np.zeros((n, 1)), np.cumsum(p, 1)], 1).clip(0, 1).astype(np.float32)
    sym = np.array([rng.choice(lp - 1, p=pi / pi.sum()) for pi in p.astype(np.float64)], dtype=np.int16)
    b, cnt = arithmetic.arithmetic_encode(torch.tensor(sym).cuda(), torch.tensor(cdf).cuda(), chunk, n, lp)
    rb, rcnt = orc.hac_encode(sym, cdf, chunk)
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    assert np.array_equal(b.cpu().numpy(), rb)          # byte-identical to the coder loop of arithmetic_kernel.cu
    dec = arithmetic.arithmetic_decode(torch.tensor(cdf).cuda(), b, cnt, chunk, n, lp).cpu().numpy()
    assert np.array_equal(dec, sym)
    assert np.array_equal(orc.hac_decode(cdf, rb, rcnt, chunk), sym)


def test_encodings_cuda_file_roundtrip(torch_cuda, tmp_path):
    torch = torch_cuda
    from gauspcc_amd import encodings_cuda as ec

    g = torch.Generator(device="cpu").manual_seed(3)
    n = 150_000 // 5
    mean = (torch.randn(n, generator=g) * 2).cuda()
    scale = (torch.rand(n, generator=g) * 3 + 0.05).cuda()
    Q = torch.full((n,), 0.5).cuda()
    x = torch.round((mean + scale * torch.randn(n, generator=g).cuda()) / Q) * Q
    bits = ec.encoder_gaussian_chunk(x, mean, scale, Q, file_name=str(tmp_path / "feat.b"), chunk_size=20000)
    assert bits > 0 and (tmp_path / "feat_0.b").exists() and (tmp_path / "feat_1.b").exists()
    y = ec.decoder_gaussian_chunk(mean, scale, Q, file_name=str(tmp_path / "feat.b"), chunk_size=20000)
    assert torch.equal(y, x)
    mask = (torch.rand(40000, generator=g) < 0.3).float().cuda()
    ec.encoder(mask, file_name=str(tmp_path / "masks.b"))
    assert torch.equal(ec.decoder(40000, file_name=str(tmp_path / "masks.b")).float(), mask)


@pytest.mark.parametrize("num_dim,n_features", [(3, 4), (2, 4), (3, 2)])
def test_gridencoder_forward_bit_exact(torch_cuda, orc, num_dim, n_features):
    torch = torch_cuda
    from gauspcc_amd.gridencoder import GridEncoder

    torch.manual_seed(num_dim * 10 + n_features)
    res = (18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514) if num_dim == 3 else (130, 258, 514, 1026)
    enc = GridEncoder(num_dim=num_dim, n_features=n_features, resolutions_list=res, log2_hashmap_size=13 if num_dim == 3 else 15, ste_binary=True).cuda()
    enc.params.data.uniform_(-1, 1)
    x = torch.rand(3000, num_dim).cuda()
    x[:4] = torch.tensor([[0.0] * num_dim, [1.0] * num_dim, [1.5] * num_dim, [0.5] * num_dim])  # borders and out of range
    out = enc(x).cpu().numpy()
    emb = np.where(enc.params.detach().cpu().numpy() >= 0, 1.0, -1.0).astype(np.float32)
    ref = orc.grid_forward(x.cpu().numpy(), emb, enc.offsets_list.cpu().numpy(), enc.resolutions_list.cpu().numpy())
    ref = ref.transpose(1, 0, 2).reshape(3000, -1)
    assert out.shape == ref.shape == (3000, len(res) * n_features)
    assert np.array_equal(out, ref)
    assert np.all(out[2] == 0)          # out-of-range input -> zeros (gridencoder.cu:135-160)


def test_gridencoder_binary_voxel_mask(torch_cuda, orc):
    torch = torch_cuda
    from gauspcc_amd.gridencoder import GridEncoder

    torch.manual_seed(5)
    enc = GridEncoder(num_dim=3, n_features=4, resolutions_list=(18, 33, 59), log2_hashmap_size=13, ste_binary=False).cuda()
    enc.params.data.uniform_(-1, 1)
    x = torch.rand(2000, 3).cuda()
    bv = (torch.rand(32, 32, 32) < 0.2).cuda()
    out = enc(x, binary_vxl=bv).cpu().numpy()
    ref = orc.grid_forward(x.cpu().numpy(), enc.params.detach().cpu().numpy(), enc.offsets_list.cpu().numpy(), enc.resolutions_list.cpu().numpy(),
                           rb=32, binary_vxl=bv.cpu().numpy().astype(np.uint8)).transpose(1, 0, 2).reshape(2000, -1)
    assert np.array_equal(out, ref)


def test_torchac_interface_bytes_match_oracle(torch_cuda, orc):
    """torchac.encode_float_cdf / encode_int16_normalized_cdf are one range-coder stream: bytes must equal the
    oracle's coder (which follows torchac's loop) on the same integerised CDF."""
    torch = torch_cuda
    from gauspcc_amd import torchac

    rng = np.random.RandomState(11)
    n, lp = 6000, 9
    p = rng.dirichlet(np.ones(lp - 1), size=n).astype(np.float32)
    cdf = np.concatenate([np.zeros((n, 1), np.float32), np.cumsum(p, 1)], 1).clip(0, 1).astype(np.float32)
    sym = np.array([rng.choice(lp - 1, p=pi / pi.sum()) for pi in p.astype(np.float64)], dtype=np.int16)
    cdf_i = orc.cdf_to_int16(cdf)
    ref = orc.rc_encode(cdf_i.view(np.uint16), sym.astype(np.uint8))
    b1 = torchac.encode_float_cdf(torch.tensor(cdf), torch.tensor(sym), check_input_bounds=True)       # CPU tensors, as torchac takes them
    b2 = torchac.encode_int16_normalized_cdf(torch.tensor(cdf_i).cuda(), torch.tensor(sym).cuda())
    assert b1 == ref and b2 == ref
    d1 = torchac.decode_float_cdf(torch.tensor(cdf), b1)
    d2 = torchac.decode_int16_normalized_cdf(torch.tensor(cdf_i), b2)
    assert d1.dtype == torch.int16 and not d1.is_cuda
    assert np.array_equal(d1.numpy(), sym) and np.array_equal(d2.numpy(), sym)
    # leading dims are flattened like torchac does
    b3 = torchac.encode_float_cdf(torch.tensor(cdf).view(60, 100, lp), torch.tensor(sym).view(60, 100))
    assert b3 == ref
    assert torchac.decode_float_cdf(torch.tensor(cdf).view(60, 100, lp), b3).shape == (60, 100)


@pytest.mark.parametrize("n,chunk", [(30000, 10000), (4097, 1000)])
def test_fused_gaussian_matches_table_path(torch_cuda, n, chunk):
    """gsac_encode_gaussian / gsac_decode_gaussian (CDF entries evaluated inside the coder) produce the bytes of
    calculate_cdf + arithmetic_encode and decode what arithmetic_decode decodes (encodings_cuda.py:336-371, 399-433)."""
    torch = torch_cuda
    from gauspcc_amd import arithmetic

    g = torch.Generator(device="cpu").manual_seed(n)
    mean = (torch.randn(n, generator=g) * 2).cuda()
    scale = (torch.rand(n, generator=g) * 3 + 0.05).cuda()
    scale[:7] = 0.0                                             # clamped to 1e-9 inside the CDF
    q = (torch.rand(n, generator=g) * 0.5 + 0.75).cuda()
    x = (mean + torch.randn(n, generator=g).cuda() * scale).contiguous()
    mn, mx, b, cnt = arithmetic.encode_gaussian(x, mean, scale, q, chunk)
    xi = torch.round(x / q)
    assert mn == float(xi.min()) and mx == float(xi.max())
    lower = arithmetic.calculate_cdf(mean, scale, q, xi.min(), xi.max())
    sym = (xi - xi.min()).to(torch.int16).contiguous()
    b2, cnt2 = arithmetic.arithmetic_encode(sym, lower, chunk, n, int(lower.shape[1]))
    assert torch.equal(cnt.cpu(), cnt2.cpu())
    assert torch.equal(b.cpu(), b2.cpu())                       # same `.b` payload without ever writing the table
    dec = arithmetic.decode_gaussian(mean, scale, q, mn, mx, b, cnt, chunk)
    assert torch.equal(dec, xi * q)
    dec2 = arithmetic.arithmetic_decode(lower, b, cnt, chunk, n, int(lower.shape[1])).to(torch.float32)
    assert torch.equal((dec2 + xi.min()) * q, dec)


def test_gaussian_slices_write_the_per_slice_files(torch_cuda, tmp_path):
    """encoder_gaussian_slices (every slice of an attribute in one device call) writes the files encoder_gaussian_chunk
    writes slice by slice -- ragged slices, one of them empty -- and decoder_gaussian_slices reads them back."""
    torch = torch_cuda
    from gauspcc_amd import encodings_cuda as ec

    g = torch.Generator(device="cpu").manual_seed(11)
    bounds = [0, 15000, 15000, 27001, 27002, 60000]           # 10000-symbol chunks: 2 / 0 / 2 / 1 / 4 chunks
    n = bounds[-1]
    mean = (torch.randn(n, generator=g) * 2).cuda()
    scale = (torch.rand(n, generator=g) * 3 + 0.05).cuda()
    q = (torch.rand(n, generator=g) * 0.5 + 0.75).cuda()
    x = (mean + torch.randn(n, generator=g).cuda() * scale * (1 + torch.arange(n).cuda() / n * 3)).contiguous()   # ranges differ per slice
    a = [str(tmp_path / f"a_{s}.b") for s in range(5)]
    b = [str(tmp_path / f"b_{s}.b") for s in range(5)]
    bits = ec.encoder_gaussian_slices(x, mean, scale, q, bounds, a)
    for s in range(5):
        lo, hi = bounds[s], bounds[s + 1]
        ref_bits = ec.encoder_gaussian_chunk(x[lo:hi], mean[lo:hi], scale[lo:hi], q[lo:hi], file_name=b[s])
        assert bits[s] == ref_bits
        if hi > lo:
            assert open(a[s].replace(".b", "_0.b"), "rb").read() == open(b[s].replace(".b", "_0.b"), "rb").read()
    dec = ec.decoder_gaussian_slices(mean, scale, q, bounds, b)         # reads the files of the slice-by-slice encoder
    assert torch.equal(dec, torch.round(x / q) * q)


def test_b_files_equal_oracle_written_files(torch_cuda, orc, tmp_path):
    """a19: the `.b` containers (HAC/utils/encodings_cuda.py:366-376, 455-464) as whole files.  Expected files are put
    together here from the ORACLE coder's output: Bernoulli `f32 p | i32 len | cnt | payload` from the exact float CDF
    [0, 1 - p, 1]; Gaussian `f32 min | f32 max | i32 len | cnt | payload` from the CDF table the device computed (erfc comes
    from two math libraries, which may round a table entry differently; the table itself is checked in
    test_calculate_cdf_matches_oracle)."""
    torch = torch_cuda
    from gauspcc_amd import arithmetic
    from gauspcc_amd import encodings_cuda as ec

    g = torch.Generator(device="cpu").manual_seed(21)
    # Bernoulli: masks-like input (n, K, 1) of {0, 1}; two chunks of 10000 and a ragged third
    x = (torch.rand(2345, 10, 1, generator=g) < 0.27).float().cuda()
    bits = ec.encoder(x, file_name=str(tmp_path / "masks.b"))
    xs = x.view(-1).cpu().numpy()
    p1 = np.float32((x.view(-1).sum() / x.numel()).item())              # the float32 the reference stores (:440, 456)
    cdf = np.empty((xs.size, 3), np.float32)
    cdf[:, 0], cdf[:, 1], cdf[:, 2] = 0.0, np.float32(1) - p1, 1.0      # :445-448: [0, 1 - p, 1] in float32
    payload, cnt = orc.hac_encode(xs.astype(np.int16), cdf, 10000)
    want = p1.tobytes() + np.array([4 * len(cnt)], np.int32).tobytes() + cnt.astype(np.int32).tobytes() + payload.tobytes()
    got = (tmp_path / "masks.b").read_bytes()
    assert got == want
    assert bits == (len(payload) + 4 * len(cnt)) * 8 + 64
    assert np.array_equal(ec.decoder(xs.size, file_name=str(tmp_path / "masks.b")).cpu().numpy(), xs.astype(np.int16))
    # Gaussian
    n = 23456
    mean = (torch.randn(n, generator=g) * 2).cuda(); scale = (torch.rand(n, generator=g) * 3 + 0.05).cuda()
    q = (torch.rand(n, generator=g) * 0.5 + 0.75).cuda()
    xq = torch.round((mean + torch.randn(n, generator=g).cuda() * scale) / q) * q
    bits = ec.encoder_gaussian(xq, mean, scale, q, file_name=str(tmp_path / "feat.b"))
    xi = torch.round(xq / q)
    mn, mx = float(xi.min()), float(xi.max())
    lower = arithmetic.calculate_cdf(mean, scale, q, mn, mx).cpu().numpy()
    payload, cnt = orc.hac_encode((xi - mn).to(torch.int16).cpu().numpy(), lower, 10000)
    want = np.float32(mn).tobytes() + np.float32(mx).tobytes() + np.array([4 * len(cnt)], np.int32).tobytes() + cnt.astype(np.int32).tobytes() + payload.tobytes()
    assert (tmp_path / "feat.b").read_bytes() == want
    assert bits == (len(payload) + 4 * len(cnt)) * 8 + 96
    assert torch.equal(ec.decoder_gaussian(mean, scale, q, file_name=str(tmp_path / "feat.b")), xq)


def test_psnr_matches_reference_golden(torch_cuda, golden_dir):
    """a21: the product's psnr (gauspcc_amd/rasterizer.py; HAC/utils/image_utils.py:17-19) on the device against the values
    the reference function gave for the same images (tests/golden/image.npz, make_golden.py)."""
    torch = torch_cuda
    from gauspcc_amd.rasterizer import psnr

    z = np.load(f"{golden_dir}/image.npz")
    got = psnr(torch.tensor(z["a"]).cuda(), torch.tensor(z["b"]).cuda())
    assert got.shape == tuple(z["psnr"].shape) and got.is_cuda
    np.testing.assert_allclose(got.cpu().numpy(), z["psnr"], rtol=1e-5)


def test_mixture_coder_matches_table_path_and_oracle_files(torch_cuda, orc, tmp_path):
    """HAC++ (HAC-plus/utils/encodings_cuda.py:177-317; callers HAC-plus/scene/gaussian_model.py:1315, 1499): the two-component
    Gaussian-mixture coder.  (1) the device's mixture table == sum of its single-Gaussian tables times the weights, clamped
    (the reference's torch expression, op for op) and within 2e-7 of the oracle's (two erfc implementations); (2) the fused
    coder writes the `.b` file the ORACLE coder assembles from that table; (3) round trip through the wrappers, scalar and
    tensor Q, chunked file names; (4) a one-component mixture with weight 1 is the plain Gaussian coder, byte for byte."""
    torch = torch_cuda
    from gauspcc_amd import arithmetic
    from gauspcc_amd import encodings_cuda as ec

    g = torch.Generator(device="cpu").manual_seed(33)
    n = 23_457
    mean = [(torch.randn(n, generator=g) * 2).cuda(), (torch.randn(n, generator=g) * 2).cuda()]
    scale = [(torch.rand(n, generator=g) * 3 + 0.05).cuda(), (torch.rand(n, generator=g) * 0.5 + 1e-3).cuda()]
    w = torch.softmax(torch.randn(n, 2, generator=g), dim=-1).cuda()
    prob = [w[:, 0].contiguous(), w[:, 1].contiguous()]
    q = (torch.rand(n, generator=g) * 0.5 + 0.75).cuda()
    pick = (torch.rand(n, generator=g) < w[:, 0].cpu()).cuda()
    x = torch.where(pick, mean[0] + torch.randn(n, generator=g).cuda() * scale[0], mean[1] + torch.randn(n, generator=g).cuda() * scale[1])
    xq = torch.round(x / q) * q
    xi = torch.round(xq / q)
    mn, mx = float(xi.min()), float(xi.max())
    # (1) the table
    table = arithmetic.calculate_cdf_mixed(mean, scale, prob, q, mn, mx)
    ref = arithmetic.calculate_cdf(mean[0], scale[0], q, mn, mx) * prob[0].unsqueeze(-1)
    ref += arithmetic.calculate_cdf(mean[1], scale[1], q, mn, mx) * prob[1].unsqueeze(-1)
    assert torch.equal(table, torch.clamp(ref, min=0.0, max=1.0))
    otab = orc.gaussian_mixed_cdf([t.cpu().numpy() for t in mean], [t.cpu().numpy() for t in scale], [t.cpu().numpy() for t in prob], q.cpu().numpy(), int(mn), int(mx))
    assert np.abs(table.cpu().numpy() - otab).max() < 2e-7
    # (2) whole file == oracle-assembled file
    bits = ec.encoder_gaussian_mixed(xq, mean, scale, prob, q, file_name=str(tmp_path / "feat.b"))
    payload, cnt = orc.hac_encode((xi - mn).to(torch.int16).cpu().numpy(), table.cpu().numpy(), 10000)
    want = np.float32(mn).tobytes() + np.float32(mx).tobytes() + np.array([4 * len(cnt)], np.int32).tobytes() + cnt.astype(np.int32).tobytes() + payload.tobytes()
    assert (tmp_path / "feat.b").read_bytes() == want
    assert bits == (len(payload) + 4 * len(cnt)) * 8 + 96
    # fused == table path on the device as well
    tb, tc = arithmetic.arithmetic_encode((xi - mn).to(torch.int16), table, 10000, n, table.shape[1])
    assert tb.cpu().numpy().tobytes() == payload.tobytes() and np.array_equal(tc.cpu().numpy(), cnt)
    # (3) round trips
    assert torch.equal(ec.decoder_gaussian_mixed(mean, scale, prob, q, file_name=str(tmp_path / "feat.b")), xq)
    xs = torch.round(x)                                               # scalar Q = 1, pieces of 10 000 elements -> feat2_0.b .. feat2_2.b
    b2 = ec.encoder_gaussian_mixed_chunk(xs, mean, scale, prob, 1, file_name=str(tmp_path / "feat2.b"), chunk_size=10_000)
    assert sorted(p.name for p in tmp_path.glob("feat2_*.b")) == ["feat2_0.b", "feat2_1.b", "feat2_2.b"] and b2 > 0
    assert torch.equal(ec.decoder_gaussian_mixed_chunk(mean, scale, prob, 1, file_name=str(tmp_path / "feat2.b"), chunk_size=10_000), xs)
    # (4) K = 1, weight 1: the single-Gaussian coder
    one = torch.ones(n, device="cuda")
    ec.encoder_gaussian_mixed(xq, mean[:1], scale[:1], [one], q, file_name=str(tmp_path / "k1.b"))
    ec.encoder_gaussian(xq, mean[0], scale[0], q, file_name=str(tmp_path / "g.b"))
    assert (tmp_path / "k1.b").read_bytes() == (tmp_path / "g.b").read_bytes()


def test_factorized_coder_roundtrip_and_table(torch_cuda, orc, tmp_path):
    """encoder_factorized / decoder_factorized (HAC/utils/encodings_cuda.py:38-175): a per-channel learned density given as
    `lower_func` (the entropy bottleneck's cumulative logits).  The payload equals the oracle coder's on the same table, the
    file layout is the Gaussian one, and the round trip is exact for every chunking."""
    torch = torch_cuda
    from gauspcc_amd import encodings_cuda as ec

    g = torch.Generator(device="cpu").manual_seed(5)
    N, Cdim, Q = 1234, 6, 0.5
    a = (torch.rand(Cdim, 1, 1, generator=g) * 1.5 + 0.3).cuda(); b = (torch.randn(Cdim, 1, 1, generator=g) * 0.7).cuda()
    lower_func = lambda v, stop_gradient=False: a * v + b               # monotone per channel, like _logits_cumulative
    x = torch.round((torch.randn(N, Cdim, generator=g).cuda() * 2 - b.view(1, -1)) / Q) * Q
    bits = ec.encoder_factorized(x, lower_func, Q, file_name=str(tmp_path / "f.b"))
    assert torch.equal(ec.decoder_factorized(lower_func, Q, N, Cdim, file_name=str(tmp_path / "f.b")), x)
    xi = torch.round(x / Q)
    mn, mx = float(xi.min()), float(xi.max())
    table = ec._factorized_table(lower_func, Q, mn, mx, Cdim, N, x.device)
    assert table.shape == (N * Cdim, int(mx - mn) + 2) and float(table.min()) >= 0 and float(table.max()) <= 1
    payload, cnt = orc.hac_encode((xi - mn).to(torch.int16).view(-1).cpu().numpy(), table.cpu().numpy(), 10000)
    want = np.float32(mn).tobytes() + np.float32(mx).tobytes() + np.array([4 * len(cnt)], np.int32).tobytes() + cnt.astype(np.int32).tobytes() + payload.tobytes()
    assert (tmp_path / "f.b").read_bytes() == want and bits == (len(payload) + 4 * len(cnt)) * 8 + 96
    ec.encoder_factorized_chunk(x, lower_func, Q, file_name=str(tmp_path / "fc.b"), chunk_size=500)
    assert len(list(tmp_path.glob("fc_*.b"))) == 3
    assert torch.equal(ec.decoder_factorized_chunk(lower_func, Q, N, Cdim, file_name=str(tmp_path / "fc.b"), chunk_size=500), x)
