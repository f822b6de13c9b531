"""The oracle checked against implementations that share no code with it (CPU only).

The arithmetic of the dominant kernels lives in packages that are absent from the reference tree (torchsparse 2.1.0,
torchac 0.9.3, diff-gaussian-rasterization), so no reference output can pin `orc_conv`, `orc_head` or
`orc_raster_forward`.  What CAN be pinned is that they compute the operators the reference's call sites name:

  * `spnn.Conv3d(C, C, k)` on a sparse tensor (network_ue_4stage_conv.py:17-62) == a dense `F.conv3d` cross-correlation
    on the densified grid, read at the occupied voxels: pins neighbour = coord + delta, the x-fastest enumeration of the
    k^3 offsets against the (k^3, Cin, Cout) weight slices, and the `in @ W[o]` orientation;
  * `Linear - ReLU - Linear - Softmax` (network_ue_4stage_conv.py:65-94) == torch.nn.functional, within the north star's
    1e-5 on probabilities; the int16 CDF == `_convert_to_int_and_normalize` of torch's own cumsum up to +-2 counts;
  * the rasteriser forward == a matrix-form splat (Sigma' = J W Sigma W^T J^T in float64, every pixel over
    every depth-sorted Gaussian, no tile lists) within 0.01 dB;
  * a reference-layout container from the C writer == what the reference's reader lines (pcc_utils.py:271-276) and the
    pinned `unpack_byte_stream` (kit/op.py:39-48, golden pack.npz) take apart, stream by stream.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def _sparse_cube(side, fill, seed):
    rng = np.random.RandomState(seed)
    occ = rng.rand(side, side, side) < fill            # [z][y][x]
    z, y, x = np.nonzero(occ)
    xyz = np.stack([x, y, z], 1).astype(np.int32)
    order = np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))   # (z, y, x) raster order, what sort_CF produces (kit/op.py:17-30)
    return xyz[order]


@pytest.mark.parametrize("k", [3, 5])
@pytest.mark.parametrize("fill,shift", [(0.3, 0), (0.05, -11), (0.9, 100)])
def test_conv_equals_dense_conv3d(orc, k, fill, shift):
    side, C = 24, 32
    xyz = _sparse_cube(side, fill, seed=10 * k + int(fill * 100))
    n = len(xyz)
    rng = np.random.RandomState(k)
    x = rng.randn(n, C).astype(np.float32)
    w = (rng.randn(k ** 3, C, C) / math.sqrt(C * k ** 3)).astype(np.float32)
    res = rng.randn(n, C).astype(np.float32)
    # the oracle: neighbour table from the (shifted, possibly negative) coordinates, then the offset-by-offset sum
    nb = orc.nbr(xyz + shift, k)
    got = orc.conv(x, nb, w)
    got_rr = orc.conv(x, nb, w, res=res, relu=True)
    # independent: densify, cross-correlate, read back.  weight[co, ci, kz, ky, kx] = W[kx + k ky + k^2 kz][ci][co]
    dense = torch.zeros(1, C, side, side, side, dtype=torch.float64)
    dense[0, :, xyz[:, 2], xyz[:, 1], xyz[:, 0]] = torch.tensor(x.T, dtype=torch.float64)
    wt = torch.tensor(w, dtype=torch.float64).reshape(k, k, k, C, C).permute(4, 3, 0, 1, 2).contiguous()
    ref = F.conv3d(dense, wt, padding=k // 2)[0][:, xyz[:, 2], xyz[:, 1], xyz[:, 0]].T.numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 1e-5 * max(scale, 1.0), (np.abs(got - ref).max(), scale)
    ref_rr = np.maximum(ref + res, 0.0)
    assert np.abs(got_rr - ref_rr).max() < 1e-5 * max(np.abs(ref_rr).max(), 1.0)
    # the number of present neighbours is what the dense occupancy says
    occ = torch.zeros(1, 1, side, side, side, dtype=torch.float64)
    occ[0, 0, xyz[:, 2], xyz[:, 1], xyz[:, 0]] = 1.0
    cnt = F.conv3d(occ, torch.ones(1, 1, k, k, k, dtype=torch.float64), padding=k // 2)[0, 0][xyz[:, 2], xyz[:, 1], xyz[:, 0]].numpy()
    assert np.array_equal((nb >= 0).sum(1), cnt.astype(np.int64))


def test_conv_offset_enumeration_is_x_fastest(orc):
    """One voxel pair, one-hot weights: the weight slice that connects a voxel to its +x neighbour is slice r + 1 (x fastest),
    to its +z neighbour slice r + k^2 ... -- the enumeration a transposed loader option has to permute."""
    k, C, r = 5, 32, 2
    centre = k ** 3 // 2
    for axis, stride in ((0, 1), (1, k), (2, k * k)):
        xyz = np.zeros((2, 3), np.int32)
        xyz[1, axis] = 1
        nb = orc.nbr(xyz, k)
        assert nb[0, centre + stride] == 1 and nb[1, centre - stride] == 0 and nb[0, centre] == 0
        x = np.zeros((2, C), np.float32); x[1, 3] = 2.0
        w = np.zeros((k ** 3, C, C), np.float32); w[centre + stride, 3, 7] = 0.5
        out = orc.conv(x, nb, w)
        assert out[0, 7] == 1.0 and np.count_nonzero(out) == 1     # out[i] = in[coord_i + delta] @ W[o(delta)]


@pytest.mark.parametrize("m", [2, 4, 16])
def test_head_equals_torch_softmax(orc, m):
    rng = np.random.RandomState(m)
    n, C = 4000, 32
    x = (rng.randn(n, C) * 2).astype(np.float32)
    w1 = (rng.randn(C, C) / math.sqrt(C)).astype(np.float32); b1 = rng.randn(C).astype(np.float32) * 0.1
    w2 = (rng.randn(m, C) / math.sqrt(C) * 3).astype(np.float32); b2 = rng.randn(m).astype(np.float32)
    prob, cdf = orc.head(x, w1, b1, w2, b2)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    ref = torch.softmax(F.linear(F.relu(F.linear(t(x), t(w1), t(b1))), t(w2), t(b2)), dim=-1)    # network_ue_4stage_conv.py:65-94
    assert np.abs(prob - ref.numpy()).max() < 1e-5                                                 # north-star tolerance on probabilities
    ref32 = torch.softmax(F.linear(F.relu(F.linear(torch.tensor(x), torch.tensor(w1), torch.tensor(b1))), torch.tensor(w2), torch.tensor(b2)), dim=-1)
    assert np.abs(prob - ref32.numpy()).max() < 1e-5
    # pcc_utils.py:146-171 on torch's own float32 probabilities, integerised by the pinned restatement of kit/op.py:50-79
    cdf_f = torch.cat([torch.zeros(n, 1), torch.cumsum(ref32, dim=-1)], dim=-1).clamp(0, 1).numpy()
    ref_int = orc.cdf_to_int16(cdf_f).view(np.uint16).astype(np.int64)
    d = np.abs(cdf.astype(np.int64) - ref_int)
    d = np.minimum(d, 65536 - d)
    assert d[:, :-1].max() <= 2, d.max()              # the last entry wraps to 0 / is never read (App. B)
    assert np.array_equal(cdf[:, 0], np.zeros(n, np.uint16))


# ------------------------------------------------------------------ rasteriser
def _camera(W, H, fovx=1.0, eye=(0.3, -0.2, -6.0), yaw=0.1):
    """World-to-camera (rotation about y + translation) and the projection of HAC/utils/graphics_utils.py:51-71
    (getProjectionMatrix, z_sign = 1), both stored transposed as HAC/scene/cameras.py:48-57 does."""
    c, s = math.cos(yaw), math.sin(yaw)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
    w2c = np.eye(4); w2c[:3, :3] = R; w2c[:3, 3] = -R @ np.array(eye, np.float64)
    fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    znear, zfar = 0.01, 100.0
    P = np.zeros((4, 4))
    P[0, 0] = 1 / tx; P[1, 1] = 1 / ty; P[3, 2] = 1.0; P[2, 2] = zfar / (zfar - znear); P[2, 3] = -(zfar * znear) / (zfar - znear)
    view_t = w2c.T.astype(np.float32)
    full_t = (w2c.T @ P.T).astype(np.float32)
    return w2c, P, view_t, full_t, tx, ty


def _naive_splat(bg, W, H, means, colors, opac, scales, rots, w2c, P, tx, ty):
    """App. E in matrix form, float64, one pixel at a time over ALL Gaussians in depth order."""
    n = len(means)
    fx, fy = W / (2 * tx), H / (2 * ty)
    ph = np.c_[means.astype(np.float64), np.ones(n)]
    pv = ph @ w2c.T
    hom = ph @ (P @ w2c).T
    ndc = hom[:, :2] / (hom[:, 3:4] + 1e-7)
    px = ((ndc[:, 0] + 1) * W - 1) / 2; py = ((ndc[:, 1] + 1) * H - 1) / 2
    q = rots.astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                  np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                  np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    S = scales.astype(np.float64)
    Sig = np.einsum("nij,nj,nkj->nik", R, S * S, R)
    Wm = w2c[:3, :3]
    items = []
    for i in range(n):
        tz = pv[i, 2]
        if tz <= 0.2:
            continue
        txx = min(1.3 * tx, max(-1.3 * tx, pv[i, 0] / tz)) * tz
        tyy = min(1.3 * ty, max(-1.3 * ty, pv[i, 1] / tz)) * tz
        J = np.array([[fx / tz, 0, -fx * txx / tz ** 2], [0, fy / tz, -fy * tyy / tz ** 2]])
        cov = J @ Wm @ Sig[i] @ Wm.T @ J.T
        cov[0, 0] += 0.3; cov[1, 1] += 0.3
        det = np.linalg.det(cov)
        if det == 0:
            continue
        mid = 0.5 * (cov[0, 0] + cov[1, 1])
        rad = math.ceil(3 * math.sqrt(mid + math.sqrt(max(0.1, mid * mid - det))))
        gx, gy = (W + 15) // 16, (H + 15) // 16
        x0 = min(gx, max(0, int((px[i] - rad) / 16))); x1 = min(gx, max(0, int((px[i] + rad + 15) / 16)))
        y0 = min(gy, max(0, int((py[i] - rad) / 16))); y1 = min(gy, max(0, int((py[i] + rad + 15) / 16)))
        if (x1 - x0) * (y1 - y0) == 0:
            continue
        items.append((np.float32(tz), i, np.linalg.inv(cov), rad, (x0, y0, x1, y1)))
    items.sort(key=lambda t: (t[0], t[1]))
    img = np.zeros((3, H, W))
    for yy in range(H):
        for xx in range(W):
            T, C = 1.0, np.zeros(3)
            for tz, i, con, rad, (x0, y0, x1, y1) in items:
                if not (x0 <= xx // 16 < x1 and y0 <= yy // 16 < y1):
                    continue
                d = np.array([px[i] - xx, py[i] - yy])
                power = -0.5 * d @ con @ d
                if power > 0:
                    continue
                alpha = min(0.99, float(opac[i, 0]) * math.exp(power))
                if alpha < 1 / 255:
                    continue
                if T * (1 - alpha) < 1e-4:
                    break
                C += colors[i] * alpha * T
                T *= 1 - alpha
            img[:, yy, xx] = C + T * bg
    radii = np.zeros(n, np.int32)
    for _, i, _, rad, _ in items:
        radii[i] = rad
    return img, radii


def test_rasteriser_equals_naive_splat(orc):
    W, H, n = 72, 40, 160
    rng = np.random.RandomState(5)
    means = ((rng.rand(n, 3) - 0.5) * np.array([7, 4, 6])).astype(np.float32)
    means[:8, 2] -= 9.0                                   # behind the camera
    scales = np.exp(rng.randn(n, 3) * 0.6 - 1.6).astype(np.float32)
    rots = rng.randn(n, 4).astype(np.float32); rots /= np.linalg.norm(rots, axis=1, keepdims=True)
    opac = (1 / (1 + np.exp(-rng.randn(n, 1) * 2))).astype(np.float32)
    colors = rng.rand(n, 3).astype(np.float32)
    bg = np.array([0.1, 0.25, 0.4], np.float32)
    w2c, P, view_t, full_t, tx, ty = _camera(W, H)
    img, radii, L = orc.raster_forward(bg, W, H, means, colors, opac, scales, 1.0, rots, view_t, full_t, tx, ty)
    ref, rradii = _naive_splat(bg, W, H, means, colors, opac, scales, rots, w2c, P, tx, ty)
    vis = rradii > 0
    assert 0.5 * n < vis.sum() < n
    # the radius is a ceil(): float32 vs float64 may differ by one count on a boundary, never by more
    assert np.array_equal(radii > 0, vis) and np.abs(radii - rradii).max() <= 1 and (radii != rradii).mean() < 0.02
    d = np.abs(img - ref)
    assert np.percentile(d, 99) < 2e-4 and d.mean() < 2e-5, (d.max(), d.mean())
    target = np.clip(ref + rng.randn(*ref.shape) * 0.05, 0, 1).astype(np.float32)
    p_orc = orc.psnr(np.clip(img, 0, 1), target).mean()
    p_ref = orc.psnr(np.clip(ref, 0, 1).astype(np.float32), target).mean()
    assert abs(p_orc - p_ref) < 0.01                     # the north star's PSNR tolerance


# ------------------------------------------------------------------ reference-layout container
@pytest.mark.parametrize("kname", ["synth_model_k5", "synth_model_k3"])
def test_v0_container_parses_with_the_pinned_reader(orc, request, kname):
    from gauspcc_amd.synth import synthetic_cloud

    model = request.getfixturevalue(kname)
    pts = synthetic_cloud(3000, seed=11, extent_log2=9) - 100       # negative coordinates too
    data = orc.encode(model, pts, chunk_log2=0, posq=2, trace=True)
    tr = orc.trace()
    levels = orc.tree_build(pts)
    # the reference's reader, pcc_utils.py:271-276
    posQ = np.frombuffer(data[:2], dtype=np.float16)[0]
    base_x_len = int(np.frombuffer(data[2:6], dtype=np.int32)[0])
    base_x_coords = np.frombuffer(data[6:6 + base_x_len * 12], dtype=np.int32).reshape(-1, 3)
    base_x_feats = np.frombuffer(data[6 + base_x_len * 12:6 + base_x_len * 13], dtype=np.uint8)
    byte_stream = data[6 + base_x_len * 13:]
    assert posQ == np.float16(2) and 0 < base_x_len < 64             # the FOG loop stops below 64 nodes (:83-89)
    assert np.array_equal(base_x_coords, levels[0][0]) and np.array_equal(base_x_feats, levels[0][1])
    streams = orc.unpack_byte_stream(byte_stream)                     # kit/op.py:39-48, pinned by golden/pack.npz
    assert len(streams) == 4 * (len(levels) - 1) == 4 * len(tr)       # level-major, stage-minor (:146-183)
    assert sum(len(s) for s in streams) + 2 + 4 * len(streams) == len(byte_stream)    # nothing behind the last stream
    for d, lv in enumerate(tr):
        assert np.array_equal(lv["xyz"], levels[d + 1][0])
        occ = levels[d + 1][1]
        syms = [(occ >> 7) & 1, (occ >> 6) & 1, (occ >> 4) & 3, occ & 15]   # :112-115
        for s in range(4):
            assert np.array_equal(lv["sym"][s], syms[s])
            stream = streams[4 * d + s]
            assert stream == orc.rc_encode(lv["cdf"][s], lv["sym"][s])       # one coder stream per (level, stage)
            assert np.array_equal(orc.rc_decode(lv["cdf"][s], stream), lv["sym"][s])
    dec, pq = orc.decode(model, data)
    assert pq == np.float16(2)
    assert np.array_equal(dec[np.lexsort((dec[:, 0], dec[:, 1], dec[:, 2]))], pts[np.lexsort((pts[:, 0], pts[:, 1], pts[:, 2]))])


def test_oracle_any_int32_position(orc, synth_model_k3):
    """Clouds outside (-2^20, 2^20) (the reference keeps int32 torchsparse coordinates, pcc_utils.py:73): a translation by a
    multiple of 2^21 >= 2^L changes the base coordinates of the container by shift >> L and nothing else; an unaligned far
    cloud round-trips; the levels of the tree are the floor-halvings of the absolute coordinates."""
    from gauspcc_amd.synth import synthetic_cloud

    pts = synthetic_cloud(3000, seed=5, extent_log2=10)
    T = np.array([3 << 28, -(1 << 30), (1 << 29) + (1 << 21)], np.int64)
    far = (pts.astype(np.int64) + T).astype(np.int32)
    a, b = orc.encode(synth_model_k3, pts, chunk_log2=0), orc.encode(synth_model_k3, far, chunk_log2=0)
    bn = int(np.frombuffer(a[2:6], np.int32)[0])
    L = (a[6 + 13 * bn] | a[7 + 13 * bn] << 8) // 4 + 1
    ba = np.frombuffer(a[6:6 + 12 * bn], np.int32).reshape(-1, 3).astype(np.int64)
    bb = np.frombuffer(b[6:6 + 12 * bn], np.int32).reshape(-1, 3).astype(np.int64)
    assert np.array_equal(bb - ba, np.tile(T >> L, (bn, 1))) and a[6 + 12 * bn:] == b[6 + 12 * bn:]
    assert np.array_equal(orc.decode(synth_model_k3, b)[0].astype(np.int64), orc.decode(synth_model_k3, a)[0].astype(np.int64) + T)
    far2 = (pts.astype(np.int64) + np.array([2 ** 31 - 5000, -2 ** 31 + 17, 123456789])).astype(np.int32)
    dec = orc.decode(synth_model_k3, orc.encode(synth_model_k3, far2, chunk_log2=10))[0]
    srt = lambda x: x[np.lexsort((x[:, 0], x[:, 1], x[:, 2]))]
    assert np.array_equal(srt(dec), srt(far2))
    levels = orc.tree_build(far2)
    cur = far2.astype(np.int64)
    for c, o in reversed(levels):                  # kit/nn.py:38-55: parents = unique(coords >> 1), occupancy = OR of octant bits
        par = cur >> 1
        u, inv = np.unique(par, axis=0, return_inverse=True)
        u = u[np.lexsort((u[:, 0], u[:, 1], u[:, 2]))]
        assert np.array_equal(u, c.astype(np.int64))
        key = {tuple(r): i for i, r in enumerate(u)}
        occ = np.zeros(len(u), np.uint8)
        for p_, ch in zip(par, cur):
            occ[key[tuple(p_)]] |= np.uint8(1 << int((ch[0] & 1) | ((ch[1] & 1) << 1) | ((ch[2] & 1) << 2)))
        assert np.array_equal(occ, o)
        cur = u
    with pytest.raises(ValueError, match="extent"):
        orc.encode(synth_model_k3, np.array([[0, 0, 0], [2 ** 21, 5, 5]], np.int32))
