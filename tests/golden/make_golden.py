"""Generate tests/golden/*.npz by IMPORTING the reference's pure-python helpers
from /root/reference (this container only; the reference never travels).

    python tests/golden/make_golden.py

Only inputs and the reference's outputs are stored -- no reference source text.
torchsparse / torchac are absent here, so they are stubbed in
sys.modules just far enough for `pcc_utils` to import; none of the captured
functions touches them.

Captured:
  morton.npz    calculate_morton_order   HAC/utils/pcc_utils.py:12-22
  sort_cf.npz   sort_CF / sort_C         src/ai_pcc/GausPcgc/kit/op.py:6-30
  cdf_int.npz   _convert_to_int_and_normalize   kit/op.py:50-79
  pack.npz      pack_byte_stream_ls / unpack_byte_stream   kit/op.py:32-48
  image.npz     psnr, getProjectionMatrix, getWorld2View2  HAC/utils/image_utils.py:17-19,
                HAC/utils/graphics_utils.py:38-71
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("torchac")
    ts = _stub("torchsparse", SparseTensor=object)
    nn = _stub("torchsparse.nn")
    fn = _stub("torchsparse.nn.functional")
    ts.nn = nn
    nn.functional = fn
    sys.path.insert(0, os.path.join(REF, "src/ai_pcc/GausPcgc"))
    sys.path.insert(0, os.path.join(REF, "src/gs_compress/HAC"))
    pcc_utils = importlib.import_module("utils.pcc_utils")
    op = importlib.import_module("kit.op")
    image_utils = importlib.import_module("utils.image_utils")
    graphics_utils = importlib.import_module("utils.graphics_utils")
    return pcc_utils, op, image_utils, graphics_utils


def main():
    pcc_utils, op, image_utils, graphics_utils = import_reference()
    rng = np.random.RandomState(20261001)

    # ---- calculate_morton_order: unique points, several shapes/dtypes/sign patterns
    cases = {}
    specs = [
        ("cube_small", 500, (0, 64, 0, 64, 0, 64), np.float32),
        ("negative", 2000, (-300, 300, -50, 80, -1000, -200), np.float32),
        ("flat_x", 1500, (5, 6, -40, 900, 0, 300), np.float32),
        ("noncubic_int", 3000, (0, 4000, 0, 17, -8, 8), np.int32),
        ("wide", 4000, (-30000, 30000, -30000, 30000, -30000, 30000), np.float64),
        ("single", 1, (3, 4, 3, 4, 3, 4), np.float32),
    ]
    for name, n, (x0, x1, y0, y1, z0, z1), dt in specs:
        pts = np.stack([rng.randint(x0, x1, 4 * n), rng.randint(y0, y1, 4 * n), rng.randint(z0, z1, 4 * n)], 1)
        pts = np.unique(pts, axis=0)
        pts = pts[rng.permutation(len(pts))[:n]]
        x = torch.tensor(pts.astype(dt))
        perm = pcc_utils.calculate_morton_order(x).numpy()
        cases[f"{name}_in"] = pts.astype(dt)
        cases[f"{name}_perm"] = perm.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "morton.npz"), **cases)

    # ---- sort_CF / sort_C
    cases = {}
    for i, n in enumerate((1, 37, 1000)):
        c = np.concatenate([np.zeros((n, 1), np.int64), rng.randint(-20, 20, (n, 3))], 1).astype(np.int32)
        c = np.unique(c, axis=0)
        c = c[rng.permutation(len(c))]
        f = rng.randn(len(c), 3).astype(np.float32)
        sc, sf = op.sort_CF(torch.tensor(c), torch.tensor(f))
        cases[f"c{i}_in"], cases[f"f{i}_in"] = c, f
        cases[f"c{i}_out"], cases[f"f{i}_out"] = sc.numpy(), sf.numpy()
        cases[f"c{i}_sortC"] = op.sort_C(torch.tensor(c)).numpy()
    np.savez_compressed(os.path.join(HERE, "sort_cf.npz"), **cases)

    # ---- _convert_to_int_and_normalize on softmax CDFs, Lp in {3,5,17}, incl. exact 0/1 and .5 ties
    cases = {}
    for lp in (3, 5, 17):
        logits = torch.tensor(rng.randn(400, lp - 1).astype(np.float32) * 3)
        p = torch.softmax(logits, -1)
        cdf = torch.cat((p[:, 0:1] * 0, p.cumsum(-1)), -1).clamp(0, 1)
        scale = 65536 - (lp - 1)
        ties = (torch.arange(0, 40, dtype=torch.float32).view(-1, 1) * 97 + 0.5) / scale  # x.5 after scaling
        ties = torch.cat([torch.zeros(40, 1), ties.repeat(1, lp - 2).cumsum(-1) if lp > 2 else ties, torch.ones(40, 1)], -1)[:, :lp].clamp(0, 1)
        edge = torch.tensor([[0.0] * (lp - 1) + [1.0], [0.0] + [1.0] * (lp - 1), [0.0] + [0.999] * (lp - 2) + [1.0]], dtype=torch.float32)
        allc = torch.cat([cdf, ties, edge], 0).contiguous()
        cases[f"lp{lp}_in"] = allc.numpy()
        cases[f"lp{lp}_out"] = op._convert_to_int_and_normalize(allc.clone(), True).numpy()
    np.savez_compressed(os.path.join(HERE, "cdf_int.npz"), **cases)

    # ---- pack / unpack
    streams = [b"ab", b"", b"xyz", bytes(rng.randint(0, 256, 300).astype(np.uint8))]
    packed = op.pack_byte_stream_ls(streams)
    un = op.unpack_byte_stream(packed)
    assert un == streams
    np.savez_compressed(
        os.path.join(HERE, "pack.npz"),
        packed=np.frombuffer(packed, np.uint8),
        **{f"s{i}": np.frombuffer(s, np.uint8) for i, s in enumerate(streams)},
    )

    # ---- psnr and camera matrices
    a = torch.tensor(rng.rand(3, 24, 32).astype(np.float32))
    b = (a + torch.tensor(rng.randn(3, 24, 32).astype(np.float32)) * 0.05).clamp(0, 1)
    ps = image_utils.psnr(a, b)
    R = np.linalg.qr(rng.randn(3, 3))[0]
    t = rng.randn(3)
    w2v = graphics_utils.getWorld2View2(R, t, np.array([0.1, -0.2, 0.3]), 1.5)
    proj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=1.1, fovY=0.8).numpy()
    np.savez_compressed(os.path.join(HERE, "image.npz"), a=a.numpy(), b=b.numpy(), psnr=ps.numpy(), R=R, t=t, w2v=w2v, proj=proj,
                        trans=np.array([0.1, -0.2, 0.3]), scale=np.array(1.5), fov=np.array([1.1, 0.8]), zplanes=np.array([0.01, 100.0]))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
