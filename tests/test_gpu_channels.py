"""GPU tests of channel counts other than 32: the reference's signatures take `channels`
(HAC/utils/pcc_utils.py:28,65 -> Network(channels, kernel_size), GausPcgc/network_ue_4stage_conv.py:12).  Widths 16 and 64 run
csrc/network_any.hip -- the same normative arithmetic as plain kernels -- and must write the oracle's bytes for that width."""
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


def _models(orc, C, k):
    from gauspcc_amd import runtime
    from gauspcc_amd.model import tensor_table
    from gauspcc_amd.synth import synthetic_state_dict

    sd = synthetic_state_dict(C, k)
    return runtime.Model(sd, C, k, 0), orc.Model(tensor_table(sd, C, k), C, k)


@pytest.mark.parametrize("C", [16, 64])
@pytest.mark.parametrize("k,n", [(3, 37), (3, 5000), (5, 37), (5, 5000), (7, 1500)])
def test_conv3d_bit_exact_other_widths(gh, orc, C, k, n):
    rng = np.random.RandomState(k * 100 + n % 97 + C)
    pts = np.unique(rng.randint(-12, 12, (n * 3, 3)), axis=0)[:n].astype(np.int32)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))]
    n = len(pts)
    x = rng.randn(n, C).astype(np.float32)
    w = (rng.rand(k ** 3, C, C).astype(np.float32) - 0.5) * 0.2
    res = rng.randn(n, C).astype(np.float32)
    nb = orc.nbr(pts, k)
    out, pairs = gh.conv3d(pts, x, w, k)
    assert pairs == int((nb >= 0).sum())
    assert np.array_equal(out, orc.conv(x, nb, w))
    out2, _ = gh.conv3d(pts, x, w, k, res=res, relu=True)
    assert np.array_equal(out2, orc.conv(x, nb, w, res=res, relu=True))


@pytest.mark.parametrize("C", [16, 64])
@pytest.mark.parametrize("m", [2, 4, 16])
def test_head_cdf_bit_exact_other_widths(gh, orc, C, m):
    rng = np.random.RandomState(m + C)
    x = (rng.randn(2051, C) * 2).astype(np.float32)
    x[:8] *= 40
    w1 = (rng.rand(C, C).astype(np.float32) - 0.5)
    b1 = (rng.rand(C).astype(np.float32) - 0.5)
    w2 = (rng.rand(m, C).astype(np.float32) - 0.5)
    b2 = (rng.rand(m).astype(np.float32) - 0.5)
    p_ref, c_ref = orc.head(x, w1, b1, w2, b2)
    p, c = gh.head_cdf(x, w1, b1, w2, b2)
    assert np.array_equal(c, c_ref) and np.array_equal(p, p_ref)


@pytest.mark.parametrize("C,k", [(16, 5), (64, 5), (16, 3), (64, 3)])
def test_codec_bitstream_identical_to_oracle_other_widths(gh, orc, C, k):
    from gauspcc_amd.synth import synthetic_cloud

    dm, om = _models(orc, C, k)
    pts = synthetic_cloud(10_000)
    for chunk_log2 in (11, 0):
        data, st = gh.encode(dm, pts, chunk_log2)
        ref = orc.encode(om, pts, chunk_log2=chunk_log2)
        assert len(data) == len(ref) and data == ref, (C, k, chunk_log2)
        dec, _, _ = gh.decode(dm, data)
        assert np.array_equal(dec, orc.decode(om, ref)[0])


def test_batch_of_scenes_at_64_channels(gh, orc):
    """The batch axis does not care about the width: two scenes through one chain of launches, bytes == solo == oracle."""
    import torch

    from gauspcc_amd.pcc_utils import _decode_batch, _encode_batch
    from gauspcc_amd.synth import synthetic_cloud

    dm, om = _models(orc, 64, 3)
    clouds = [synthetic_cloud(6_000, seed=5), synthetic_cloud(3_000, seed=6)]
    blobs, _, batched = _encode_batch([torch.tensor(c, device=gh.dev()) for c in clouds], dm, 11, [1, 1])
    assert batched
    for b, c in zip(blobs, clouds):
        assert b == orc.encode(om, c, chunk_log2=11)
    outs, _, _, _ = _decode_batch(blobs, dm, gh.dev())
    for o, b in zip(outs, blobs):
        assert np.array_equal(o.cpu().numpy(), orc.decode(om, b)[0])


def test_plugin_api_takes_channels(gh, tmp_path):
    import torch

    from gauspcc_amd.pcc_utils import compress_point_cloud, decompress_point_cloud
    from gauspcc_amd.synth import synthetic_cloud

    pts = synthetic_cloud(4_000, seed=9)
    out = compress_point_cloud(torch.tensor(pts), "synthetic", str(tmp_path / "a.bin"), channels=16, kernel_size=3)
    dec = decompress_point_cloud(out["output_path"], "synthetic", channels=16, kernel_size=3)
    got = dec["point_cloud"].cpu().numpy().astype(np.int64)
    assert np.array_equal(got[np.lexsort((got[:, 0], got[:, 1], got[:, 2]))], pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))])
