"""GPU tests of the HAC++ attribute loop ("what's missing" 1): conduct_encoding -> conduct_decoding of
src/gs_compress/HAC-plus/scene/gaussian_model.py:1209-1395 / 1396-1590 on a model object that exposes what the reference's
GaussianModel exposes -- `feat` in five ten-channel groups under a two-component mixture whose second component comes from the
channel-context MLP on the groups already coded."""
import os

import numpy as np
import pytest

from tests.test_gpu_hac_codec import _Model, torch_cuda  # noqa: F401  (the HAC stub: geometry, hash grids, accessors)

pytestmark = pytest.mark.gpu


def _ctx_mlp(torch, g, dev, tiny=False):
    """Channel_CTX_fea (HAC-plus/scene/gaussian_model.py:117-168) restated: five MLPs Linear(150 + 10 c, 40) - LeakyReLU - Linear(40, 30)
    on cat([d0 .. d(c-1), mean_scale]); forward(fea_q, mean_scale, to_dec=c) returns group c's (mean, scale, prob) adjustments."""

    class ChannelCtx(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for c in range(5):
                setattr(self, f"MLP_d{c}", torch.nn.Sequential(torch.nn.Linear(50 * 3 + 10 * c, 40), torch.nn.LeakyReLU(inplace=True), torch.nn.Linear(40, 30)))

        def forward(self, fea_q, mean_scale, to_dec=-1):
            d = torch.split(fea_q, [10] * 5, dim=-1)
            outs = [torch.chunk(getattr(self, f"MLP_d{c}")(torch.cat(list(d[:c]) + [mean_scale], dim=-1)), chunks=3, dim=-1) for c in range(5)]
            if 0 <= to_dec < 5:
                return outs[to_dec]
            return tuple(torch.cat([o[i] for o in outs], dim=-1) for i in range(3))

    m = ChannelCtx()
    for p in m.parameters():
        p.data = torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else 1.0 / np.sqrt(p.shape[-1]))
    return m.to(dev)


class _ModelPlus(_Model):
    """What HAC++ adds to the slice of GaussianModel the codec touches: mlp_grid with the extra `prob` head, the channel-context MLP,
    one more mask slot per anchor, get_mask_anchor of shape (N, 1) (:465-476)."""

    def __init__(self, torch, n, seed=0):
        super().__init__(torch, n, seed=seed)
        g = torch.Generator(device="cpu").manual_seed(seed + 1000)
        dev = torch.device("cuda", 0)
        fd, K = self.feat_dim, self.n_offsets
        self.mlp_grid = torch.nn.Sequential(torch.nn.Linear(self.encoding_xyz.output_dim, fd * 2), torch.nn.ReLU(True),
                                            torch.nn.Linear(fd * 2, (fd + 6 + 3 * K) * 2 + fd + 1 + 1 + 1)).to(dev)   # (:370-374)
        self.mlp_deform = _ctx_mlp(torch, g, dev)
        self._mask = torch.cat([self._mask, torch.zeros(self._mask.shape[0], 1, 1, device=dev)], dim=1)               # (N, K + 1, 1)

    @property
    def get_mask(self):
        import torch
        if self.decoded_version:
            return self._mask[:, :self.n_offsets, :]
        s = torch.sigmoid(self._mask[:, :self.n_offsets, :])
        return ((s > 0.01).float() - s).detach() + s

    @property
    def get_mask_anchor(self):
        import torch
        rate = torch.mean(self.get_mask, dim=1)
        return ((rate > 0.0).float() - rate).detach() + rate          # (N, 1)

    @property
    def get_deform_mlp(self):
        return self.mlp_deform


@pytest.mark.parametrize("n,din,dh,dout", [(3000, 150, 40, 30), (4099, 190, 40, 30), (777, 48, 100, 225), (5000, 48, 100, 195), (33, 20, 30, 30)])
def test_mlp2_leaky_matches_oracle(torch_cuda, orc, n, din, dh, dout):
    """gshac_mlp2_act (LeakyReLU between the layers): the channel-context MLPs and HAC++'s mlp_grid run in padded classes of the
    matrix-pipe kernel -- still the oracle's chain (specified fp32 order), and the MLP torch computes."""
    torch = torch_cuda
    from gauspcc_amd import hac_codec, hac_plus_codec

    rng = np.random.RandomState(n + din)
    x = rng.randn(n, din).astype(np.float32)
    w1 = (rng.randn(dh, din) / np.sqrt(din)).astype(np.float32); b1 = rng.randn(dh).astype(np.float32) * 0.1
    w2 = (rng.randn(dout, dh) / np.sqrt(dh)).astype(np.float32); b2 = rng.randn(dout).astype(np.float32) * 0.1
    t = [torch.tensor(a).cuda() for a in (x, w1, b1, w2, b2)]
    y = hac_plus_codec.mlp2_act(*t, 0.01).cpu().numpy()
    assert np.array_equal(y, orc.mlp2(x, w1, b1, w2, b2, slope=0.01))
    ref = torch.nn.functional.leaky_relu(torch.tensor(x) @ torch.tensor(w1).T + torch.tensor(b1), 0.01) @ torch.tensor(w2).T + torch.tensor(b2)
    np.testing.assert_allclose(y, ref.numpy(), rtol=2e-5, atol=2e-5)
    assert np.array_equal(hac_codec.mlp2(*t).cpu().numpy(), orc.mlp2(x, w1, b1, w2, b2))       # and ReLU through the same classes


def test_conduct_encoding_decoding_roundtrip_hac_plus(torch_cuda, orc, tmp_path):
    torch = torch_cuda
    from gauspcc_amd import encodings_cuda, hac_plus_codec

    enc = _ModelPlus(torch, 7000, seed=5)
    log = hac_plus_codec.conduct_encoding(enc, str(tmp_path), ckpt_path="synthetic")
    assert "Encoded sizes in MB" in log and "EncTime" in log
    # the reference's per-component time line (HAC-plus/scene/gaussian_model.py:1384-1392): the same seven fields, `anchor` = order + compress_point_cloud
    import re
    m = re.search(r"\nEncoded time in s: anchor ([0-9.e-]+), feat ([0-9.e-]+), scaling ([0-9.e-]+), offsets ([0-9.e-]+), hash ([0-9.e-]+), masks ([0-9.e-]+), Total ([0-9.e-]+)$", log)
    assert m, log
    t = [float(x) for x in m.groups()]
    assert all(x >= 0 for x in t) and t[0] > 0 and t[1] > 0 and sum(t[:6]) <= t[6] * 1.05 + 1e-3
    keep = enc.get_mask_anchor.to(torch.bool)[:, 0]
    n, mb = int(keep.sum()), 3000
    steps = -(-n // mb)
    files = set(os.listdir(tmp_path))
    want = {"xyz_pcc.bin", "hash.b", "masks.b", "x_bound_min.pkl", "x_bound_max.pkl"} | {f"{a}_{s}_0.b" for a in ("scaling", "offsets") for s in range(steps)} \
        | {f"feat_{s}_{c}_0.b" for s in range(steps) for c in range(5)}                     # (:1272, :1321)
    assert want <= files, sorted(want - files)

    dec = _ModelPlus(torch, 10, seed=99)
    dec.encoding_xyz, dec.mlp_grid, dec.mlp_deform = enc.encoding_xyz, enc.mlp_grid, enc.mlp_deform
    dec._anchor_feat = torch.zeros(1, enc.feat_dim, device="cuda")
    msg = hac_plus_codec.conduct_decoding(dec, str(tmp_path), ckpt_path="synthetic")
    assert msg.startswith("\nDecTime")
    m = re.search(r"\nDecoded time in s: anchor ([0-9.e-]+), feat ([0-9.e-]+), scaling ([0-9.e-]+), offsets ([0-9.e-]+), hash ([0-9.e-]+), masks ([0-9.e-]+), Total ([0-9.e-]+)$", msg)   # (:1576-1584)
    assert m, msg
    t = [float(x) for x in m.groups()]
    assert t[0] > 0 and t[1] > 0 and sum(t[:6]) <= t[6] * 1.05 + 1e-3
    assert torch.equal(dec.x_bound_min, enc.x_bound_min)

    # What the decoder must reproduce, computed with plain torch the way the reference's loop does (:1262-1321): the quantised
    # attributes.  Nothing below touches hac_plus_codec.
    a_int = torch.round(enc.get_anchor[keep] / enc.voxel_size)
    key = (a_int - a_int.min(dim=0, keepdim=True).values).to(torch.int64)
    M = key.max() + 1
    order = torch.argsort(key[:, 0] + key[:, 1] * M + key[:, 2] * M * M)
    anchor = a_int[order] * enc.voxel_size
    assert torch.equal(dec._anchor.data, anchor)
    _feat, _scaling, _mask, _offs = enc._anchor_feat[keep][order], enc.get_scaling[keep][order], enc.get_mask[keep][order], enc._offset[keep][order]
    assert dec._mask.shape == (n, enc.n_offsets + 1, 1) and torch.equal(dec._mask.data[:, :enc.n_offsets], _mask)

    def ste(x, Q, mean):
        x = torch.clamp(x, min=(mean - 15_000 * Q), max=(mean + 15_000 * Q))
        return torch.round(x / Q) * Q

    def exact(got, want, Q):
        assert torch.equal(got, want), (int((got != want).sum()), float((got - want).abs().max()))

    def close(got, want, Q):
        d = (got - want).abs()
        bad = d > 2e-5 * (1 + want.abs())
        assert float(bad.float().mean()) <= 5e-4, float(bad.float().mean())
        assert bool((d[bad] <= Q[bad] * 1.001).all())

    gm = enc.get_grid_mlp
    w1, b1, w2, b2 = (t.detach().cpu().numpy() for t in (gm[0].weight, gm[0].bias, gm[2].weight, gm[2].bias))

    def oracle_mlp(x):
        return torch.tensor(orc.mlp2(x.cpu().numpy(), w1, b1, w2, b2), device=x.device)

    # Twice.  (1) EXACT: mlp_grid evaluated by the ORACLE's bit-specified Linear-ReLU-Linear chain on the host (== gshac_mlp2, the chain the codec
    # runs: test_gpu_hac_codec.py::test_mlp2_matches_oracle_bit_for_bit), the de-quantised attributes compared with torch.equal.
    # (2) APPROXIMATE by construction: mlp_grid as torch evaluates the nn.Sequential (the reference's own call) -- torch's GEMM agrees with the chain
    # to ~1e-6 and is not reproducible to the ulp from run to run, so a step Q may differ in its last bits and a value on a rounding boundary of
    # x / Q may land one step away: within 2e-5 relative, at most 5e-4 of the values off, each by exactly one step.
    fd, K = enc.feat_dim, enc.n_offsets
    with torch.no_grad():
        for grid_mlp, check in ((oracle_mlp, exact), (gm, close)):
            for s0 in range(0, n, mb):
                sl = slice(s0, min(s0 + mb, n))
                out = grid_mlp(enc.calc_interp_feat(anchor[sl]))
                mean, scale, prob, mean_s, scale_s, mean_o, scale_o, qf, qs, qo = torch.split(out, [fd, fd, fd, 6, 6, 3 * K, 3 * K, 1, 1, 1], dim=-1)
                Qf = (1 * (1 + torch.tanh(qf.contiguous()))).repeat(1, fd)
                Qs = (0.001 * (1 + torch.tanh(qs.contiguous()))).repeat(1, 6)
                Qo = (0.2 * (1 + torch.tanh(qo.contiguous()))).repeat(1, 3 * K)
                check(dec._anchor_feat.data[sl], ste(_feat[sl], Qf, enc._anchor_feat.mean()), Qf)
                check(dec._scaling.data[sl], ste(_scaling[sl], Qs, enc.get_scaling.mean()), Qs)
                m3 = _mask[sl].repeat(1, 1, 3).view(-1, 3 * K)
                check(dec._offset.data[sl].reshape(-1, 3 * K), ste(_offs[sl].reshape(-1, 3 * K), Qo, enc._offset.mean()) * m3, Qo)

    # The per-slice files of a group are the files the one-slice coder writes for the same tensors (that coder is pinned to the
    # reference's table path + arithmetic_encode in test_gpu_attributes.py): slice 1, groups 0 and 3, from the codec's own context
    # -- the channel context of group 3 is the DECODED groups 0-2, which the decoder reproduced above.
    c = hac_plus_codec._context(enc, anchor)
    feat_q = dec._anchor_feat.data
    s = 1
    rows = slice(s * mb, min((s + 1) * mb, n))
    for cc in (0, 3):
        means, scales, probs, q = hac_plus_codec._group_mixture(enc, c, feat_q, cc)
        el = slice(rows.start * 10, rows.stop * 10)
        x = feat_q[rows, cc * 10:cc * 10 + 10].contiguous().view(-1)
        ref_name = str(tmp_path / f"ref_{cc}.b")
        encodings_cuda.encoder_gaussian_mixed_chunk(x, [m[el] for m in means], [t[el] for t in scales], [p[el] for p in probs], q[el], file_name=ref_name,
                                                    chunk_size=50_0000)
        assert (tmp_path / f"ref_{cc}_0.b").read_bytes() == (tmp_path / f"feat_{s}_{cc}_0.b").read_bytes()


def test_hac_plus_mixture_beats_single_gaussian_when_context_helps(torch_cuda, tmp_path):
    """A sanity check on the wiring of the autoregressive chain: when group c is (nearly) a copy of group c - 1 and the
    channel-context MLP is built to predict exactly that, the mixture's second component must make group c almost free."""
    torch = torch_cuda
    from gauspcc_amd import hac_plus_codec

    enc = _ModelPlus(torch, 4000, seed=7)
    with torch.no_grad():
        f = enc._anchor_feat
        f[:, 10:20] = f[:, 0:10]                                     # group 1 repeats group 0
        m = enc.mlp_deform.MLP_d1                                    # input: cat([d0, mean_scale]) -> hidden 40 -> (mean, scale, prob) x 10
        for p in m.parameters():
            p.zero_()
        for j in range(10):
            m[0].weight[j, j] = 1.0; m[0].weight[10 + j, j] = -1.0   # hidden j = d0_j, hidden 10 + j = -d0_j (LeakyReLU: |.| parts)
            m[2].weight[j, j] = 1.0; m[2].weight[j, 10 + j] = -1.0 + 0.01 * 0.0
            m[2].bias[10 + j] = 1e-3                                 # scale_adj: tiny
            m[2].bias[20 + j] = 12.0                                 # prob_adj: the softmax picks component 1
        m[2].weight[:10, 10:20] *= 1.0 / (1.0 - 0.0)
    hac_plus_codec.conduct_encoding(enc, str(tmp_path), ckpt_path="synthetic")
    size = lambda c: sum(os.path.getsize(tmp_path / f) for f in os.listdir(tmp_path) if f.startswith("feat_") and f.endswith(f"_{c}_0.b"))
    # LeakyReLU(x) - LeakyReLU(-x) = 1.01 x: the prediction is within 1 % of the value; group 1 must cost a fraction of group 0
    assert size(1) < 0.6 * size(0), (size(0), size(1))
    dec = _ModelPlus(torch, 10, seed=99)
    dec.encoding_xyz, dec.mlp_grid, dec.mlp_deform = enc.encoding_xyz, enc.mlp_grid, enc.mlp_deform
    dec._anchor_feat = torch.zeros(1, enc.feat_dim, device="cuda")
    hac_plus_codec.conduct_decoding(dec, str(tmp_path), ckpt_path="synthetic")
    assert torch.equal(dec._anchor_feat.data[:, 10:20], dec._anchor_feat.data[:, 0:10])


def test_generate_neural_gaussians_on_an_undecoded_hac_plus_model(torch_cuda):
    """an un-decoded HAC++ model is quantised through the TEN-way split of its mlp_grid (mean, scale, prob, ...;
    HAC-plus/gaussian_renderer/__init__.py:119-136) -- HAC's nine-way split would take the step sizes from the right columns only by luck
    of the layout (they are the last three either way) but mis-split everything in front; the test pins the step sizes AND the quantised
    attributes against the reference's tensor program, and the Gaussians against the decoded-model path on the quantised attributes."""
    torch = torch_cuda
    import types

    from gauspcc_amd import neural_gaussians as ng
    from gauspcc_amd.hac_codec import grid_mlp

    pc = _ModelPlus(torch, 6000, seed=11)
    pc.decoded_version = False
    dev = torch.device("cuda", 0)
    F, K = pc.feat_dim, pc.n_offsets
    nn = torch.nn
    torch.manual_seed(3)
    pc.use_feat_bank = False
    pc.get_opacity_mlp = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, K), nn.Tanh()).to(dev)
    pc.get_cov_mlp = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, 7 * K)).to(dev)
    pc.get_color_mlp = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, 3 * K), nn.Sigmoid()).to(dev)
    pc.rotation_activation = torch.nn.functional.normalize
    cam = types.SimpleNamespace(camera_center=torch.tensor([0.3, -4.0, 1.1], device=dev))
    vis = torch.rand(pc.get_anchor.shape[0], device=dev) > 0.1
    with torch.no_grad():
        anchor = pc.get_anchor[vis]
        # the reference's lines (:119-136), plain torch on the same context-model output
        out = grid_mlp(pc, pc.calc_interp_feat(anchor))
        mean, scale, prob, mean_scaling, scale_scaling, mean_offsets, scale_offsets, qf, qs, qo = torch.split(
            out, split_size_or_sections=[F, F, F, 6, 6, 3 * K, 3 * K, 1, 1, 1], dim=-1)
        Q_feat = 1 * (1 + torch.tanh(qf.contiguous().repeat(1, F)))
        Q_scaling = 0.001 * (1 + torch.tanh(qs.contiguous().repeat(1, 6)))
        Q_offsets = (0.2 * (1 + torch.tanh(qo.contiguous().repeat(1, 3 * K)))).view(-1, K, 3)

        def ste(x, Q, mean):
            x = torch.clamp(x, min=(mean - 15_000 * Q), max=(mean + 15_000 * Q))
            return torch.round(x / Q) * Q

        feat_q = ste(pc._anchor_feat[vis], Q_feat, pc._anchor_feat.mean())
        scal_q = ste(pc.get_scaling[vis], Q_scaling, pc.get_scaling.mean())
        offs_q = ste(pc._offset[vis], Q_offsets, pc._offset.mean())
        gq = ng.quant_steps(pc, anchor)
        assert torch.equal(gq[0], Q_feat) and torch.equal(gq[1], Q_scaling) and torch.equal(gq[2].view(-1, K, 3), Q_offsets)
        # the Gaussians of the un-decoded model == the Gaussians of a decoded model that holds the quantised attributes
        dq = types.SimpleNamespace(feat_dim=F, n_offsets=K, decoded_version=True, use_feat_bank=False, get_anchor=anchor, _anchor_feat=feat_q, _offset=offs_q,
                                   get_scaling=scal_q, get_mask=pc.get_mask[vis], get_opacity_mlp=pc.get_opacity_mlp, get_cov_mlp=pc.get_cov_mlp,
                                   get_color_mlp=pc.get_color_mlp, rotation_activation=pc.rotation_activation)
        a = ng.generate_neural_gaussians(cam, pc, vis)
        b = ng.generate_neural_gaussians(cam, dq, None)
    assert a[5] > 0 and b[5] == 0                       # time_sub: only the un-decoded model pays for the context model
    assert a[0].shape == b[0].shape and a[0].shape[0] > 1000
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)
    # HAC++'s mask-after-the-opacity-test (:188-205) keeps exactly the offsets with opacity > 0 AND binary mask 1
    with torch.no_grad():
        ob = anchor - cam.camera_center
        od = ob.norm(dim=1, keepdim=True)
        x = torch.cat([feat_q, ob / od, od], dim=1)
        op = pc.get_opacity_mlp(x).reshape(-1, 1)
        keep = ((op > 0.0).view(-1)) & (pc.get_mask[vis].reshape(-1) > 0)
    assert abs(int(keep.sum()) - a[0].shape[0]) <= int((op.abs() < 1e-4).sum())
