"""GPU test of gsnn_generate / generate_neural_gaussians against the same operations written
in PyTorch fp32 (a floating-point kernel: torch reference + tolerance; src/gs_compress/HAC/gaussian_renderer/__init__.py:116-171)."""
import types

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    return torch


def _model(torch, n, F, K, bank, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device("cuda", 0)
    nn = torch.nn
    pc = types.SimpleNamespace()
    pc.feat_dim, pc.n_offsets, pc.decoded_version, pc.use_feat_bank = F, K, True, bank
    pc.get_anchor = (torch.rand(n, 3, generator=g) * 4 - 2).to(dev)
    pc._anchor_feat = (torch.randn(n, F, generator=g) * 0.8).to(dev)
    pc._offset = (torch.randn(n, K, 3, generator=g) * 0.3).to(dev)
    pc.get_scaling = torch.exp(torch.randn(n, 6, generator=g) * 0.4 - 2.5).to(dev)
    pc.get_mask = (torch.rand(n, K, 1, generator=g) > 0.35).float().to(dev)
    pc.get_mask_anchor = (pc.get_mask.sum(dim=1)[:, 0] > 0)
    pc.rotation_activation = torch.nn.functional.normalize
    torch.manual_seed(seed)
    pc.get_opacity_mlp = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, K), nn.Tanh()).to(dev)
    pc.get_cov_mlp = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, 7 * K)).to(dev)
    pc.get_color_mlp = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, 3 * K), nn.Sigmoid()).to(dev)
    if bank:
        pc.get_featurebank_mlp = nn.Sequential(nn.Linear(4, F), nn.ReLU(True), nn.Linear(F, 3), nn.Softmax(dim=1)).to(dev)
    cam = types.SimpleNamespace(camera_center=torch.tensor([0.3, -4.0, 1.1], device=dev))
    return pc, cam


def _torch_reference(torch, cam, pc, visible_mask):
    """The reference's tensor program (:116-171), inference path with decoded attributes."""
    anchor = pc.get_anchor[visible_mask]
    feat = pc._anchor_feat[visible_mask]
    grid_offsets = pc._offset[visible_mask]
    grid_scaling = pc.get_scaling[visible_mask]
    masks = pc.get_mask[visible_mask]
    K = pc.n_offsets
    ob_view = anchor - cam.camera_center
    ob_dist = ob_view.norm(dim=1, keepdim=True)
    ob_view = ob_view / ob_dist
    if pc.use_feat_bank:
        bank_weight = pc.get_featurebank_mlp(torch.cat([ob_view, ob_dist], dim=1)).unsqueeze(dim=1)
        f = feat.unsqueeze(dim=-1)
        f = f[:, ::4, :1].repeat([1, 4, 1]) * bank_weight[:, :, :1] + f[:, ::2, :1].repeat([1, 2, 1]) * bank_weight[:, :, 1:2] + f[:, ::1, :1] * bank_weight[:, :, 2:]
        feat = f.squeeze(dim=-1)
    x = torch.cat([feat, ob_view, ob_dist], dim=1)
    neural_opacity = pc.get_opacity_mlp(x).reshape([-1, 1]) * masks.view(-1, 1)
    mask = (neural_opacity > 0.0).view(-1)
    opacity = neural_opacity[mask]
    color = pc.get_color_mlp(x).reshape([anchor.shape[0] * K, 3])
    scale_rot = pc.get_cov_mlp(x).reshape([anchor.shape[0] * K, 7])
    offsets = grid_offsets.view([-1, 3])
    rep = torch.cat([grid_scaling, anchor], dim=-1).repeat_interleave(K, dim=0)
    allc = torch.cat([rep, color, scale_rot, offsets], dim=-1)[mask]
    scaling_repeat, repeat_anchor, color, scale_rot, offsets = allc.split([6, 3, 3, 7, 3], dim=-1)
    scaling = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])
    rot = pc.rotation_activation(scale_rot[:, 3:7])
    xyz = repeat_anchor + offsets * scaling_repeat[:, :3]
    return xyz, color, opacity, scaling, rot, neural_opacity.view(-1)


@pytest.mark.parametrize("F,K,bank,n", [(50, 10, False, 20011), (32, 10, True, 7001), (32, 5, False, 300), (32, 12, True, 9001), (32, 16, True, 4099), (50, 16, False, 15),
                                        (50, 17, False, 2000)])   # n_offsets > 16: the one-lane-per-anchor kernel
def test_generate_neural_gaussians_matches_torch(torch_cuda, F, K, bank, n):
    torch = torch_cuda
    from gauspcc_amd.neural_gaussians import generate_neural_gaussians

    pc, cam = _model(torch, n, F, K, bank, seed=F + K)
    vis = torch.rand(n, device="cuda") > 0.2
    with torch.no_grad():
        rx, rc, ro, rs, rr, nopa = _torch_reference(torch, cam, pc, vis)
    xyz, color, opacity, scaling, rot, time_sub = generate_neural_gaussians(cam, pc, vis)
    assert time_sub == 0
    # Gaussians whose neural opacity is within rounding of zero may fall on either side of the `> 0` test; every other one
    # must line up one to one.  Rows are paired through their positions (every candidate has its own anchor + offset), rows
    # without a partner must be borderline, and the values of the paired rows are compared ALWAYS.
    EPS = 1e-4
    w = torch.tensor([0.6180339, 1.3247179, 0.7548777], device="cuda", dtype=torch.float64)
    kd, kr = xyz.double() @ w, rx.double() @ w
    sr, ir = torch.sort(kr)
    pos = torch.searchsorted(sr, kd).clamp(1, len(sr) - 1)
    near = torch.where((kd - sr[pos - 1]).abs() <= (sr[pos] - kd).abs(), pos - 1, pos)
    hit = (sr[near] - kd).abs() < 1e-4
    pd, pr = torch.nonzero(hit).view(-1), ir[near[hit]]
    assert len(torch.unique(pr)) == len(pr)                                       # one partner each
    assert torch.equal(pr, torch.sort(pr).values)                                 # and the same (anchor-major) order
    lone_d = torch.ones(xyz.shape[0], dtype=torch.bool, device="cuda"); lone_d[pd] = False
    lone_r = torch.ones(rx.shape[0], dtype=torch.bool, device="cuda"); lone_r[pr] = False
    assert bool((opacity.view(-1)[lone_d] < EPS).all()) and bool((ro.view(-1)[lone_r] < EPS).all())
    assert int(lone_d.sum()) + int(lone_r.sum()) <= int(((nopa.abs() < EPS) & (nopa != 0)).sum())
    assert len(pd) > n                                                            # plenty of Gaussians survive with these masks
    for a, b, tol in ((xyz, rx, 2e-5), (color, rc, 2e-5), (opacity, ro, 2e-5), (scaling, rs, 2e-5), (rot, rr, 5e-5)):
        a, b = a[pd], b[pr]
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol, float((a - b).abs().max())
    assert opacity.min() > 0


@pytest.mark.parametrize("mask_kind", ["ones", "binary", "sparse"])
def test_without_a_visible_mask_and_with_few_survivors(torch_cuda, mask_kind):
    """visible_mask=None (the kernels index the model's rows directly) with masks that leave whole lane groups of a wave without a survivor: the
    shape that exposed a compiler-materialised lane mask being reused under a different EXEC (csrc/neural_gaussians.hip: the ROWS template parameter)."""
    torch = torch_cuda
    from gauspcc_amd.neural_gaussians import generate_neural_gaussians

    n, F, K = 5396, 50, 10
    pc, cam = _model(torch, n, F, K, False, seed=77)
    if mask_kind == "ones":
        pc.get_mask = torch.ones_like(pc.get_mask)
    elif mask_kind == "sparse":
        pc.get_mask = (torch.rand(n, K, 1, device="cuda") > 0.97).float()
    vis = torch.ones(n, dtype=torch.bool, device="cuda")
    with torch.no_grad():
        rx, rc, ro, rs, rr, nopa = _torch_reference(torch, cam, pc, vis)
    xyz, color, opacity, scaling, rot, _ = generate_neural_gaussians(cam, pc, None)
    borderline = int((nopa.abs() < 1e-4).sum())
    assert abs(xyz.shape[0] - rx.shape[0]) <= borderline
    if xyz.shape[0] == rx.shape[0]:
        for a, b, tol in ((xyz, rx, 2e-5), (color, rc, 2e-5), (opacity, ro, 2e-5), (scaling, rs, 2e-5), (rot, rr, 5e-5)):
            assert float((a - b).abs().max()) <= tol
