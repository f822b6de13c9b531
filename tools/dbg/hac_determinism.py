#!/usr/bin/env python3
"""Developer probe: is the context model deterministic run to run -- ours (hash grid + gshac_mlp2) and torch's (nn.Sequential on the same input)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_hac_codec import _Model
from gauspcc_amd import hac_codec
enc = _Model(torch, 7000, seed=5)
anchor = enc.get_anchor
ours, theirs = [], []
for rep in range(30):
    with torch.no_grad():
        feat = enc.calc_interp_feat(anchor)
        ours.append(hac_codec.grid_mlp(enc, feat).clone())
        theirs.append(enc.get_grid_mlp(feat).clone())
print("ours identical over 30 runs:", all(torch.equal(ours[0], o) for o in ours))
print("torch identical over 30 runs:", all(torch.equal(theirs[0], o) for o in theirs))
print("max |ours - torch|:", float((ours[0] - theirs[0]).abs().max()))
# slices of 3000 as the test calls torch
with torch.no_grad():
    sl = torch.cat([enc.get_grid_mlp(enc.calc_interp_feat(anchor[s:s + 3000])) for s in range(0, anchor.shape[0], 3000)])
print("torch whole == torch by slices:", torch.equal(sl, theirs[0]), float((sl - theirs[0]).abs().max()))
