"""The checkpoint path of compress_point_cloud / decompress_point_cloud (pcc_utils.py:65-67, 266-268): a torch file holding
`net.state_dict()` with the upstream key names, and the loader options for the two unverifiable
conventions of torchsparse's (k^3, Cin, Cout) kernels.  CPU part: file loading, key handling, layouts."""
import numpy as np
import pytest
import torch

from gauspcc_amd.model import conv_offset_layout, load_state_dict, tensor_table
from gauspcc_amd.synth import CONV_KEYS, synthetic_state_dict

UPSTREAM_KEYS = (["prior_embedding.weight", "target_embedding.target_res_embedding.weight", "fog.conv.kernel"]
                 + [f"prior_resnet.{i}" for i in ("0.kernel", "2.conv0.kernel", "2.conv1.kernel", "3.conv0.kernel", "3.conv1.kernel")]
                 + [f"target_resnet.{i}" for i in ("0.kernel", "2.conv0.kernel", "2.conv1.kernel", "3.conv0.kernel", "3.conv1.kernel")]
                 + [f"spatial_conv_s{s}.{j}.kernel" for s in range(4) for j in (0, 2)]
                 + [f"pred_head_s{s}.{j}.{w}" for s in range(4) for j in (0, 2) for w in ("weight", "bias")]
                 + [f"pred_head_s{s}_emb.weight" for s in (1, 2, 3)])


def _save_pt(path, sd, prefix="", wrap=False):
    t = {prefix + k: torch.tensor(v) for k, v in sd.items()}
    torch.save({"state_dict": t, "step": 500} if wrap else t, path)


def test_synthetic_state_dict_has_the_upstream_key_set():
    sd = synthetic_state_dict(32, 5)
    assert sorted(sd) == sorted(UPSTREAM_KEYS)                 # network_ue_4stage_conv.py:15-98, kit/nn.py:14-16,31,106
    assert sd["prior_resnet.0.kernel"].shape == (125, 32, 32) and sd["fog.conv.kernel"].shape == (8, 1, 1)
    assert set(CONV_KEYS) <= set(UPSTREAM_KEYS) and len(CONV_KEYS) == 18


@pytest.mark.parametrize("prefix,wrap", [("", False), ("module.", False), ("", True)])
def test_torch_checkpoint_round_trips_through_the_loader(tmp_path, prefix, wrap):
    sd = synthetic_state_dict(32, 3, seed=11)
    p = tmp_path / "best_model_ue_4stage_conv.pt"
    _save_pt(p, sd, prefix, wrap)
    want = tensor_table(sd, 32, 3)
    got = tensor_table(load_state_dict(str(p), 32, 3), 32, 3)
    assert len(got) == 39 and all(np.array_equal(a, b) and a.dtype == np.float32 for a, b in zip(got, want))
    np.savez(tmp_path / "w.npz", **sd)
    got = tensor_table(load_state_dict(str(tmp_path / "w.npz"), 32, 3), 32, 3)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_loader_reports_missing_and_misshapen_tensors(tmp_path):
    sd = synthetic_state_dict(32, 3)
    bad = dict(sd); del bad["pred_head_s2.2.bias"]
    with pytest.raises(KeyError, match="pred_head_s2.2.bias"):
        tensor_table(bad, 32, 3)
    with pytest.raises(ValueError, match="prior_resnet.0.kernel"):
        tensor_table(sd, 32, 5)                                # a k = 3 checkpoint loaded as k = 5


@pytest.mark.parametrize("k", [3, 5])
def test_offset_layout_options(orc, k):
    """offset_order='zyx' is the (k,k,k) axis transpose, flip_offsets the mirrored enumeration: a kernel stored in either
    convention and loaded with the matching option convolves like the x-fastest kernel it stands for."""
    rng = np.random.RandomState(k)
    K, C, r = k ** 3, 32, k // 2
    w = rng.randn(K, C, C).astype(np.float32)
    xyz = np.unique(rng.randint(0, 9, (400, 3)), axis=0).astype(np.int32)
    xyz = xyz[np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))]
    x = rng.randn(len(xyz), C).astype(np.float32)
    nb = orc.nbr(xyz, k)
    ref = orc.conv(x, nb, w)
    # the same kernel written down z-fastest: slice (dz+r) + k (dy+r) + k^2 (dx+r) holds W[delta]
    o = np.arange(K); dx, dy, dz = o % k, (o // k) % k, o // (k * k)
    w_zyx = np.empty_like(w); w_zyx[dz + k * dy + k * k * dx] = w[o]
    assert np.array_equal(conv_offset_layout(w_zyx, k, offset_order="zyx"), w)
    assert np.array_equal(orc.conv(x, nb, conv_offset_layout(w_zyx, k, offset_order="zyx")), ref)
    assert not np.array_equal(orc.conv(x, nb, w_zyx), ref)
    # ... and with slices attached to -delta
    w_neg = np.empty_like(w); w_neg[(2 * r - dx) + k * (2 * r - dy) + k * k * (2 * r - dz)] = w[o]
    assert np.array_equal(conv_offset_layout(w_neg, k, flip_offsets=True), w)
    both = np.empty_like(w); both[(2 * r - dz) + k * (2 * r - dy) + k * k * (2 * r - dx)] = w[o]
    assert np.array_equal(conv_offset_layout(both, k, flip_offsets=True, offset_order="zyx"), w)
    with pytest.raises(ValueError):
        conv_offset_layout(w, k, offset_order="yxz")


@pytest.mark.gpu
def test_compress_point_cloud_loads_a_torch_checkpoint(tmp_path, monkeypatch):
    """ckpt_path=<file saved by torch.save(net.state_dict())> through the plugin API: same bitstream as the same weights handed
    over as a dict; a z-fastest copy of the checkpoint gives that bitstream with GAUSPCC_OFFSET_ORDER=zyx and another without."""
    from gauspcc_amd.pcc_utils import compress_point_cloud, decompress_point_cloud
    from gauspcc_amd.synth import synthetic_cloud

    k = 3
    sd = synthetic_state_dict(32, k, seed=5)
    p = tmp_path / "best_model_ue_4stage_conv.pt"
    _save_pt(p, sd, "module.")
    pts = torch.tensor(synthetic_cloud(20_000, seed=4)).cuda()
    a = compress_point_cloud(pts, str(p), str(tmp_path / "a.bin"), kernel_size=k)
    b = compress_point_cloud(pts, sd, str(tmp_path / "b.bin"), kernel_size=k)
    assert a["file_size_bits"] == b["file_size_bits"] and (tmp_path / "a.bin").read_bytes() == (tmp_path / "b.bin").read_bytes()
    out = decompress_point_cloud(str(tmp_path / "a.bin"), str(p), kernel_size=k)
    assert out["num_points"] == 20_000
    got = out["point_cloud"].int()
    srt = lambda t: t[np.lexsort((t[:, 0], t[:, 1], t[:, 2]))]
    assert np.array_equal(srt(got.cpu().numpy()), srt(pts.cpu().numpy()))
    zsd = dict(sd)
    for key in CONV_KEYS:
        zsd[key] = np.ascontiguousarray(sd[key].reshape(k, k, k, 32, 32).transpose(2, 1, 0, 3, 4).reshape(k ** 3, 32, 32))
    pz = tmp_path / "zfast.pt"
    _save_pt(pz, zsd)
    c = compress_point_cloud(pts, str(pz), str(tmp_path / "c.bin"), kernel_size=k)
    assert (tmp_path / "c.bin").read_bytes() != (tmp_path / "a.bin").read_bytes()
    monkeypatch.setenv("GAUSPCC_OFFSET_ORDER", "zyx")
    d = compress_point_cloud(pts, str(pz), str(tmp_path / "d.bin"), kernel_size=k)
    assert (tmp_path / "d.bin").read_bytes() == (tmp_path / "a.bin").read_bytes() and d["bpp"] == a["bpp"] and c["num_points"] == 20_000


class _Marker:
    """Module-level class: a checkpoint that pickles one of these is not 'tensors only'."""


def test_checkpoint_with_pickled_objects_is_refused_unless_opted_in(tmp_path, monkeypatch):
    """the loader must not fall back to full unpickling by itself (that executes code in the file)."""
    sd = synthetic_state_dict(32, 3, seed=11)
    p = tmp_path / "objs.pt"
    t = {k: torch.tensor(v) for k, v in sd.items()}
    t["extra"] = _Marker()
    torch.save(t, p)
    monkeypatch.delenv("GAUSPCC_UNSAFE_CKPT", raising=False)
    with pytest.raises(ValueError, match="GAUSPCC_UNSAFE_CKPT"):
        load_state_dict(str(p), 32, 3)
    monkeypatch.setenv("GAUSPCC_UNSAFE_CKPT", "1")
    got = load_state_dict(str(p), 32, 3)
    assert isinstance(got["extra"], _Marker) and len(tensor_table(got, 32, 3)) == 39
