"""Deterministic synthetic inputs: anchor clouds and GausPcgc weights.

The reference ships neither data nor the trained checkpoint (README.md:73-77), so
the benchmark and the parity tests run on seeded synthetic inputs.  Everything here
is a counter-based generator (splitmix64) so that the same seed gives the same
bytes on every box, independent of numpy's Generator stream.

Cloud recipe: cluster centres uniform in a 2^16 cube,
per-cluster sigma in U(4, 40) voxels, anisotropy (1, 1, 0.15), integer-rounded,
de-duplicated, first N after a seeded shuffle.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(ctr: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (ctr + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


class CounterRNG:
    """uniform()/normal() streams addressed by (seed, stream id, index)."""

    def __init__(self, seed: int):
        self.seed = np.uint64(seed)

    def bits(self, stream: int, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            base = _splitmix64(np.array([self.seed ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))], dtype=np.uint64))[0]
            ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        return _splitmix64(ctr)

    def uniform(self, stream: int, n: int) -> np.ndarray:
        """float64 in [0, 1) with 53 random bits."""
        return (self.bits(stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, stream: int, n: int) -> np.ndarray:
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(stream * 2 + 1, m)
        u2 = self.uniform(stream * 2 + 2, m)
        r = np.sqrt(-2.0 * np.log(u1))
        out = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return out[:n]


def synthetic_cloud(n_points: int, seed: int = 1234, negative: bool = False, extent_log2: int = 16) -> np.ndarray:
    """(n_points, 3) int32, duplicate-free, 'globally sparse, locally dense'."""
    rng = CounterRNG(seed)
    n_clusters = max(4, n_points // 250)
    ext = 1 << extent_log2
    margin = ext // 32
    centres = margin + rng.uniform(1, n_clusters * 3).reshape(n_clusters, 3) * (ext - 2 * margin)
    sigma = 4.0 + 36.0 * rng.uniform(2, n_clusters)
    # random axis permutation per cluster for the thin direction
    thin = (rng.uniform(3, n_clusters) * 3).astype(np.int64)
    pts = np.zeros((0, 3), dtype=np.int64)
    want = n_points
    attempt = 0
    while pts.shape[0] < n_points:
        m = int(want * 1.6) + 64
        cid = (rng.uniform(10 + attempt * 4, m) * n_clusters).astype(np.int64)
        g = rng.normal(11 + attempt * 4, m * 3).reshape(m, 3)
        aniso = np.ones((m, 3))
        aniso[np.arange(m), thin[cid]] = 0.15
        p = np.rint(centres[cid] + g * sigma[cid, None] * aniso).astype(np.int64)
        p = np.clip(p, 0, ext - 1)
        pts = np.concatenate([pts, p], axis=0)
        # de-duplicate keeping first occurrence
        key = (pts[:, 2] << 42) | (pts[:, 1] << 21) | pts[:, 0]
        _, first = np.unique(key, return_index=True)
        pts = pts[np.sort(first)]
        want = n_points - pts.shape[0]
        attempt += 1
    # seeded shuffle, first N
    order = np.argsort(rng.bits(99, pts.shape[0]), kind="stable")
    pts = pts[order[:n_points]]
    if negative:
        pts = pts - (ext // 2)
    return np.ascontiguousarray(pts.astype(np.int32))


def solid_cloud(n_points: int, seed: int = 4321, extent_log2: int = 12) -> np.ndarray:
    """(n_points, 3) int32, duplicate-free: every voxel inside a union of seeded balls -- a SOLID cloud.  Interior octree nodes have all
    eight children (occupancy 255), so the marginal symbol distributions are sharply peaked (2.5 bits per coded node at 1 M points, 3.6
    at 200 k, instead of the 5.2 of the sparse bench cloud) and there are only ~0.16 coded nodes per point: with
    peaky_state_dict(gain=1, freq=stage_symbol_frequencies(cloud)) the codec runs at 0.4 - 0.6 bits per point, the range coder in its
    low-entropy regime (long carry runs, a few bytes per chunk).  (gain = 1: every node of a solid level sees all 125 taps, so the
    nominal initialiser already keeps the activations O(1); the sparse clouds' gain = 4 makes the logits explode here.)"""
    rng = CounterRNG(seed)
    ext = 1 << extent_log2
    n_balls = 12
    # radii so that the balls' volumes add up to ~1.25 n (overlaps and the trim below take the rest)
    rel = 0.6 + 0.8 * rng.uniform(1, n_balls)
    r = rel * (1.25 * n_points / (4.18879 * np.sum(rel ** 3))) ** (1.0 / 3.0)
    margin = int(np.ceil(r.max())) + 2
    if ext <= 2 * margin + 2:
        raise ValueError("solid_cloud: extent too small for this many points")
    ctr = margin + rng.uniform(2, n_balls * 3).reshape(n_balls, 3) * (ext - 2 * margin)
    keys = []
    for b in range(n_balls):
        R = int(np.ceil(r[b]))
        c = np.rint(ctr[b]).astype(np.int64)
        g = np.arange(-R, R + 1, dtype=np.int64)
        z, y, x = np.meshgrid(g, g, g, indexing="ij")
        inside = x * x + y * y + z * z <= r[b] * r[b]
        keys.append(((z[inside] + c[2]) << 42) | ((y[inside] + c[1]) << 21) | (x[inside] + c[0]))
    key = np.unique(np.concatenate(keys))
    if key.shape[0] < n_points:
        raise ValueError("solid_cloud: the balls hold fewer voxels than requested")
    key = key[:n_points]                 # raster-order trim: removes whole slabs of the last balls, punches no holes
    m = (1 << 21) - 1
    return np.ascontiguousarray(np.stack([key & m, (key >> 21) & m, key >> 42], axis=1).astype(np.int32))


# --------------------------------------------------------------------------- weights
STAGE_M = (2, 2, 4, 16)

CONV_KEYS = (
    [f"prior_resnet.{k}" for k in ("0.kernel", "2.conv0.kernel", "2.conv1.kernel", "3.conv0.kernel", "3.conv1.kernel")]
    + [f"target_resnet.{k}" for k in ("0.kernel", "2.conv0.kernel", "2.conv1.kernel", "3.conv0.kernel", "3.conv1.kernel")]
    + [f"spatial_conv_s{s}.{i}.kernel" for s in range(4) for i in (0, 2)]
)


def synthetic_state_dict(channels: int = 32, kernel_size: int = 5, seed: int = 7, gain: float = 4.0) -> dict:
    """Seeded weights under the upstream state-dict key names
    (network_ue_4stage_conv.py:15-98, kit/nn.py:14-16,31,106).

    Initialisers follow the reference modules: sparse conv U(+-1/sqrt(Cin*k^3)),
    nn.Linear U(+-1/sqrt(fan_in)), nn.Embedding N(0,1).  `gain` multiplies the conv
    kernels: a submanifold conv only sees ~10 of the k^3 taps, so the nominal
    initialiser makes activations vanish after a few layers; gain=4 keeps them O(1)
    so that the predicted probabilities actually vary between nodes.  It has no
    effect on the work done.
    """
    rng = CounterRNG(seed)
    C, K = channels, kernel_size ** 3
    sd = {}
    stream = [100]

    def uni(shape, bound):
        stream[0] += 1
        n = int(np.prod(shape))
        return ((rng.uniform(stream[0], n) * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)

    def nrm(shape):
        stream[0] += 1
        n = int(np.prod(shape))
        return rng.normal(stream[0], n).astype(np.float32).reshape(shape)

    sd["prior_embedding.weight"] = nrm((256, C))
    for key in CONV_KEYS:
        sd[key] = uni((K, C, C), gain / np.sqrt(C * K))
    sd["target_embedding.target_res_embedding.weight"] = nrm((8, C))
    for s, m in enumerate(STAGE_M):
        b = 1.0 / np.sqrt(C)
        sd[f"pred_head_s{s}.0.weight"] = uni((C, C), b)
        sd[f"pred_head_s{s}.0.bias"] = uni((C,), b)
        sd[f"pred_head_s{s}.2.weight"] = uni((m, C), b)
        sd[f"pred_head_s{s}.2.bias"] = uni((m,), b)
        if s > 0:
            sd[f"pred_head_s{s}_emb.weight"] = nrm((2 ** (1, 2, 4)[s - 1], C))
    sd["fog.conv.kernel"] = np.ones((8, 1, 1), dtype=np.float32)
    return sd


def stage_symbol_frequencies(points: np.ndarray):
    """Empirical frequencies of the four stage symbols over every coded octree node of a cloud (closed form: no network).
    A node's occupancy byte o = sum over children of 2^(x%2 + 2(y%2) + 4(z%2)) (kit/nn.py:38-55) is coded as four symbols
    o>>7&1, o>>6&1, o>>4&3, o&15 (pcc_utils.py:118-142); levels are halved until fewer than 64 nodes remain, as the codec does
    (pcc_utils.py:83-89).  Returns four float64 arrays of 2 / 2 / 4 / 16 frequencies."""
    c = np.unique(np.asarray(points, dtype=np.int64), axis=0)
    counts = [np.zeros(m, dtype=np.int64) for m in STAGE_M]
    while c.shape[0] >= 64:
        par = c >> 1
        bit = (c[:, 0] & 1) + 2 * (c[:, 1] & 1) + 4 * (c[:, 2] & 1)
        up, inv = np.unique(par, axis=0, return_inverse=True)
        occ = np.zeros(up.shape[0], dtype=np.int64)
        np.add.at(occ, inv.reshape(-1), np.int64(1) << bit)
        for s, sym in enumerate((occ >> 7 & 1, occ >> 6 & 1, occ >> 4 & 3, occ & 15)):
            counts[s] += np.bincount(sym, minlength=STAGE_M[s])
        c = up
    return [cn / max(1, cn.sum()) for cn in counts]


# stage_symbol_frequencies(synthetic_cloud(1_000_000, seed=1234)), rounded to 4 digits: the "peaky" model's head biases are the logs
# of these, so that the model is the same on every box without the 1 M cloud having to be rebuilt (tests check the table against the function)
PEAKY_FREQ_S1M = (
    (0.829, 0.171),
    (0.8287, 0.1713),
    (0.6867, 0.1422, 0.1419, 0.0292),
    (0.4359, 0.1192, 0.1192, 0.0125, 0.1191, 0.0127, 0.006, 0.004, 0.1195, 0.0061, 0.0124, 0.004, 0.0126, 0.004, 0.0039, 0.0088),
)   # marginal entropy 0.66 + 0.66 + 1.32 + 2.58 = 5.22 bits per coded node (x 2.7 coded nodes per point of that cloud = 14.1 bpp)


def peaky_state_dict(channels: int = 32, kernel_size: int = 5, seed: int = 7, gain: float = 4.0, freq=None, head_scale: float = 0.25) -> dict:
    """A low-rate operating point without training: the seeded weights of synthetic_state_dict with the
    LAST layer of every prediction head (network_ue_4stage_conv.py:65-94) re-targeted -- weights x head_scale, bias = log of the
    stage's empirical symbol frequencies (PEAKY_FREQ_S1M unless `freq` is given).  The predicted distributions are then the
    marginals of the occupancy symbols, modulated by the context network: peaked rows (most nodes of a sparse cloud have one child),
    long carry runs and few bytes per chunk in the range coder -- the regime a trained GausPcgc checkpoint works in."""
    sd = synthetic_state_dict(channels, kernel_size, seed, gain)
    fr = PEAKY_FREQ_S1M if freq is None else freq
    for s, m in enumerate(STAGE_M):
        f = np.maximum(np.asarray(fr[s], dtype=np.float64), 1e-6)
        sd[f"pred_head_s{s}.2.weight"] = (sd[f"pred_head_s{s}.2.weight"] * np.float32(head_scale)).astype(np.float32)
        sd[f"pred_head_s{s}.2.bias"] = np.log(f / f.sum()).astype(np.float32)
    return sd


class SyntheticGaussianModel:
    """The slice of HAC's GaussianModel that conduct_encoding / conduct_decoding / generate_neural_gaussians touch
    (src/gs_compress/HAC/scene/gaussian_model.py:111-430), filled with seeded random tensors and randomly initialised
    MLPs -- the reference ships no trained scene.  Anchors come from synthetic_cloud (voxel units x voxel_size)."""

    def __init__(self, n_anchors, feat_dim=50, n_offsets=10, seed=0, voxel_size=0.001, use_feat_bank=False, device="cuda:0"):
        import torch
        from .gridencoder import mix_3D2D_encoding

        nn = torch.nn
        g = torch.Generator(device="cpu").manual_seed(seed)
        dev = torch.device(device)
        self.feat_dim, self.n_offsets, self.voxel_size = feat_dim, n_offsets, voxel_size
        self.decoded_version, self.ste_binary, self.use_2D, self.n_features_per_level = False, True, True, 2
        self.use_feat_bank = use_feat_bank
        vox = torch.tensor(synthetic_cloud(n_anchors, seed=seed + 1, extent_log2=12)).float()
        vox = vox - vox.mean(dim=0).round()
        self._anchor = (vox * voxel_size).to(dev)
        n = self._anchor.shape[0]
        self._anchor_feat = (torch.randn(n, feat_dim, generator=g) * 0.7).to(dev)
        self._offset = (torch.randn(n, n_offsets, 3, generator=g) * 0.3).to(dev)
        self._scaling = (torch.randn(n, 6, generator=g) * 0.5 - 4.5).to(dev)
        self._mask = (torch.randn(n, n_offsets, 1, generator=g) * 4.0).to(dev)   # logits: some anchors end up fully masked
        lo, hi = self._anchor.min(dim=0).values, self._anchor.max(dim=0).values
        pad = (hi - lo) * 0.05 + 1e-3
        self.x_bound_min, self.x_bound_max = (lo - pad).view(1, 3), (hi + pad).view(1, 3)
        self.encoding_xyz = mix_3D2D_encoding(n_features=2, resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514),
                                              log2_hashmap_size=13, resolutions_list_2D=(130, 258, 514, 1026), log2_hashmap_size_2D=15,
                                              ste_binary=True, ste_multistep=False, add_noise=False, Q=1).to(dev)
        for p in self.encoding_xyz.parameters():
            p.data = torch.randn(p.shape, generator=g).to(dev)                   # only the signs matter (STE_binary)
        torch.manual_seed(seed)
        F, K = feat_dim, n_offsets
        self.mlp_grid = nn.Sequential(nn.Linear(self.encoding_xyz.output_dim, F * 2), nn.ReLU(True), nn.Linear(F * 2, (F + 6 + 3 * K) * 2 + 3)).to(dev)
        self.mlp_opacity = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, K), nn.Tanh()).to(dev)
        self.mlp_cov = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, 7 * K)).to(dev)
        self.mlp_color = nn.Sequential(nn.Linear(F + 4, F), nn.ReLU(True), nn.Linear(F, 3 * K), nn.Sigmoid()).to(dev)
        if use_feat_bank:
            self.mlp_feature_bank = nn.Sequential(nn.Linear(4, F), nn.ReLU(True), nn.Linear(F, 3), nn.Softmax(dim=1)).to(dev)
        self.rotation_activation = torch.nn.functional.normalize

    # accessors as in the reference (:347-405)
    @property
    def get_scaling(self):
        import torch
        return self._scaling if self.decoded_version else 1.0 * torch.exp(self._scaling)

    @property
    def get_mask(self):
        import torch
        if self.decoded_version:
            return self._mask
        s = torch.sigmoid(self._mask)
        return ((s > 0.01).float() - s).detach() + s

    @property
    def get_mask_anchor(self):
        import torch
        return (torch.sum(self.get_mask, dim=1)[:, 0]) > 0

    @property
    def get_anchor(self):
        import torch
        return self._anchor if self.decoded_version else torch.round(self._anchor / self.voxel_size) * self.voxel_size

    get_grid_mlp = property(lambda self: self.mlp_grid)
    get_opacity_mlp = property(lambda self: self.mlp_opacity)
    get_cov_mlp = property(lambda self: self.mlp_cov)
    get_color_mlp = property(lambda self: self.mlp_color)
    get_featurebank_mlp = property(lambda self: self.mlp_feature_bank)

    def get_encoding_params(self):
        import torch
        e = self.encoding_xyz
        p = torch.cat([e.encoding_xyz.params, e.encoding_xy.params, e.encoding_xz.params, e.encoding_yz.params], dim=0)
        return (p >= 0) * (+1.0) + (p < 0) * (-1.0)                              # STE_binary (:283-286)

    def calc_interp_feat(self, x):
        return self.encoding_xyz((x - self.x_bound_min) / (self.x_bound_max - self.x_bound_min))


class SyntheticGaussianModelPlus(SyntheticGaussianModel):
    """What HAC++ adds to the slice of GaussianModel its codec touches (src/gs_compress/HAC-plus/scene/gaussian_model.py): `mlp_grid`
    with the extra `prob` head and two more adjustments (:370-374), the channel-context MLP `Channel_CTX_fea` behind `get_deform_mlp`
    (:117-168, :377: five MLPs Linear(150 + 10 c, 40) - LeakyReLU - Linear(40, 30) on the groups already coded + mean_scale), one more
    mask slot per anchor and `get_mask_anchor` of shape (N, 1) (:465-476).  Seeded random weights, as the base class."""

    def __init__(self, n_anchors, seed=0, **kw):
        import numpy as np
        import torch

        super().__init__(n_anchors, seed=seed, **kw)
        nn = torch.nn
        dev = self._anchor.device
        g = torch.Generator(device="cpu").manual_seed(seed + 1000)
        F, K = self.feat_dim, self.n_offsets
        torch.manual_seed(seed + 1)
        self.mlp_grid = nn.Sequential(nn.Linear(self.encoding_xyz.output_dim, F * 2), nn.ReLU(True), nn.Linear(F * 2, (F + 6 + 3 * K) * 2 + F + 1 + 1 + 1)).to(dev)

        class ChannelCtx(nn.Module):
            def __init__(self):
                super().__init__()
                for c in range(5):
                    setattr(self, f"MLP_d{c}", nn.Sequential(nn.Linear(F * 3 + 10 * c, 40), nn.LeakyReLU(inplace=True), nn.Linear(40, 30)))

            def forward(self, fea_q, mean_scale, to_dec=-1):
                d = torch.split(fea_q, [10] * 5, dim=-1)
                outs = [torch.chunk(getattr(self, f"MLP_d{c}")(torch.cat(list(d[:c]) + [mean_scale], dim=-1)), chunks=3, dim=-1) for c in range(5)]
                if 0 <= to_dec < 5:
                    return outs[to_dec]
                return tuple(torch.cat([o[i] for o in outs], dim=-1) for i in range(3))

        m = ChannelCtx()
        for p in m.parameters():
            p.data = torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else 1.0 / float(np.sqrt(p.shape[-1])))
        self.mlp_deform = m.to(dev)
        self._mask = torch.cat([self._mask, torch.zeros(self._mask.shape[0], 1, 1, device=dev)], dim=1)   # (N, K + 1, 1)

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

    get_deform_mlp = property(lambda self: self.mlp_deform)
