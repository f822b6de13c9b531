"""Forward-only mirror of HAC's hash-grid encoder (src/gs_compress/HAC/utils/encodings.py:92-311,
HAC/submodules/gridencoder.zip): `grid_encode` (the forward of _grid_encode), `GridEncoder`
and `mix_3D2D_encoding` (HAC/scene/gaussian_model.py:43-109) with the reference's constructor
arguments, buffers and output layout.  Inference (encode / decode time) only: no autograd.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, runtime


def grid_encode(inputs, embeddings, offsets_list, resolutions_list, calc_grad_inputs=False, min_level_id=None, n_levels_calc=1, binary_vxl=None, PV=0):
    """encodings.py:95-170 (forward).  Returns [N, n_levels_calc * n_features]."""
    if calc_grad_inputs:
        raise NotImplementedError("gauspcc_amd.gridencoder is forward-only (dy_dx / backward are training-side)")
    inputs = inputs.contiguous().float()
    Rb = 128
    bv = None
    if binary_vxl is not None:
        binary_vxl = binary_vxl.contiguous()
        Rb = binary_vxl.shape[-1]
        assert len(binary_vxl.shape) == inputs.shape[-1]
        bv = binary_vxl.to(torch.uint8).contiguous()
    N, num_dim = inputs.shape
    n_features = embeddings.shape[1]
    embeddings = embeddings.contiguous().float()
    outputs = torch.empty(n_levels_calc, N, n_features, device=inputs.device, dtype=torch.float32)
    if isinstance(min_level_id, int) or min_level_id is None:
        lo = int(min_level_id or 0)
        off = offsets_list[lo:lo + n_levels_calc + 1].to(torch.int32).contiguous()
        res = resolutions_list[lo:lo + n_levels_calc].to(torch.int32).contiguous()
        ml = None
    else:
        off = offsets_list.to(torch.int32).contiguous()
        res = resolutions_list.to(torch.int32).contiguous()
        ml = min_level_id.to(torch.int32).contiguous()
    _lib.check(_lib.lib().gsge_forward(runtime.context(inputs.device), inputs.data_ptr(), embeddings.data_ptr(), off.data_ptr(), res.data_ptr(),
                                       outputs.data_ptr(), N, num_dim, n_features, n_levels_calc, Rb,
                                       None if bv is None else bv.data_ptr(), None if ml is None else ml.data_ptr(),
                                       runtime.stream_ptr(inputs.device)))
    return outputs.permute(1, 0, 2).reshape(N, n_levels_calc * n_features)


class GridEncoder(nn.Module):
    def __init__(self, num_dim=3, n_features=2, resolutions_list=(16, 23, 32, 46, 64, 92, 128, 184, 256, 368, 512, 736),
                 log2_hashmap_size=19, ste_binary=True, ste_multistep=False, add_noise=False, Q=1):
        super().__init__()
        resolutions_list = torch.tensor(resolutions_list).to(torch.int)
        n_levels = resolutions_list.numel()
        self.num_dim, self.n_levels, self.n_features = num_dim, n_levels, n_features
        self.log2_hashmap_size = log2_hashmap_size
        self.output_dim = n_levels * n_features
        self.ste_binary, self.ste_multistep, self.add_noise, self.Q = ste_binary, ste_multistep, add_noise, Q
        offsets_list, offset = [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(n_levels):
            resolution = resolutions_list[i].item()
            params_in_level = min(self.max_params, resolution ** num_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets_list.append(offset)
            offset += params_in_level
        offsets_list.append(offset)
        self.register_buffer('offsets_list', torch.from_numpy(np.array(offsets_list, dtype=np.int32)))
        self.register_buffer('resolutions_list', resolutions_list)
        self.n_params = offsets_list[-1] * n_features
        self.params = nn.Parameter(torch.empty(offset, n_features))
        self.reset_parameters()
        self.n_output_dims = n_levels * n_features

    def reset_parameters(self):
        std = 1e-4
        self.params.data.uniform_(-std, std)

    @torch.no_grad()
    def forward(self, inputs, min_level_id=None, max_level_id=None, test_phase=False, outspace_params=None, binary_vxl=None, PV=0):
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.num_dim)
        params = outspace_params if outspace_params is not None else self.params
        if self.ste_binary:
            embeddings = (params >= 0) * (+1.0) + (params < 0) * (-1.0)       # STE_binary.forward (encodings.py:25-33)
        elif self.add_noise and not test_phase:
            embeddings = params + (torch.rand_like(params) - 0.5) * (1 / self.Q)
        elif self.ste_multistep or (self.add_noise and test_phase):
            embeddings = torch.round(params / self.Q) * self.Q                # STE_multistep.forward
        else:
            embeddings = params
        min_level_id = 0 if min_level_id is None else max(min_level_id, 0)
        max_level_id = self.n_levels if max_level_id is None else min(max_level_id, self.n_levels)
        n_levels_calc = max_level_id - min_level_id
        outputs = grid_encode(inputs, embeddings, self.offsets_list, self.resolutions_list, False, min_level_id, n_levels_calc, binary_vxl, PV)
        return outputs.view(prefix_shape + [n_levels_calc * self.n_features])


class mix_3D2D_encoding(nn.Module):
    """HAC/scene/gaussian_model.py:43-109."""

    def __init__(self, n_features, resolutions_list, log2_hashmap_size, resolutions_list_2D, log2_hashmap_size_2D,
                 ste_binary, ste_multistep, add_noise, Q):
        super().__init__()
        kw = dict(n_features=n_features, ste_binary=ste_binary, ste_multistep=ste_multistep, add_noise=add_noise, Q=Q)
        self.encoding_xyz = GridEncoder(num_dim=3, resolutions_list=resolutions_list, log2_hashmap_size=log2_hashmap_size, **kw)
        self.encoding_xy = GridEncoder(num_dim=2, resolutions_list=resolutions_list_2D, log2_hashmap_size=log2_hashmap_size_2D, **kw)
        self.encoding_xz = GridEncoder(num_dim=2, resolutions_list=resolutions_list_2D, log2_hashmap_size=log2_hashmap_size_2D, **kw)
        self.encoding_yz = GridEncoder(num_dim=2, resolutions_list=resolutions_list_2D, log2_hashmap_size=log2_hashmap_size_2D, **kw)
        self.output_dim = self.encoding_xyz.output_dim + self.encoding_xy.output_dim + self.encoding_xz.output_dim + self.encoding_yz.output_dim

    def forward(self, x):
        x_x, y_y, z_z = torch.chunk(x, 3, dim=-1)
        out_xyz = self.encoding_xyz(x)
        out_xy = self.encoding_xy(torch.cat([x_x, y_y], dim=-1))
        out_xz = self.encoding_xz(torch.cat([x_x, z_z], dim=-1))
        out_yz = self.encoding_yz(torch.cat([y_y, z_z], dim=-1))
        return torch.cat([out_xyz, out_xy, out_xz, out_yz], dim=-1)
