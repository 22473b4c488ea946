"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32, autograd) restatement of the reference's attribute decode.

  triplane_features : HexPlaneField.forward -> interpolate_ms_features -> grid_sample_wrapper
                      (sings/rec/models/modules/hexplane.py:46-105,161-190): normalise with the aabb, per scale the product
                      over the planes (0,1), (0,2), (1,2) of F.grid_sample(bilinear, border, align_corners=True), concat.
  geometry_decoder / appearance_decoder : modules/decoders.py:16-110, functional form on a state_dict.
Pinned by tests/golden/decode_golden.npz (tests/golden/gen_decode_golden.py runs the reference modules themselves).
Only tests/ may import this module.
"""
import itertools

import torch
import torch.nn.functional as F


def triplane_features(pts, grids, aabb):
    """grids: list over scales of 3 tensors [1,F,H,W]; aabb [2,3]."""
    p = (pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
    outs = []
    for planes in grids:
        interp_space = 1.
        for ci, comb in enumerate(itertools.combinations(range(3), 2)):
            coords = p[..., comb].view(1, 1, -1, 2)
            interp = F.grid_sample(planes[ci], coords, align_corners=True, mode='bilinear', padding_mode='border')
            interp_space = interp_space * interp.view(planes[ci].shape[1], -1).t()
        outs.append(interp_space)
    return torch.cat(outs, dim=-1)


def _lin(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def geometry_decoder(x, sd, isotropic=True):
    h = F.gelu(_lin(F.gelu(_lin(x, sd, "net.0")), sd, "net.2"))
    xyz_offsets = _lin(h, sd, "xyz_offsets")
    rotations = _lin(h, sd, "rotations.0") if not isotropic else None
    scales_aux = _lin(F.gelu(_lin(h, sd, "scales.0")), sd, "scales.2")
    scales = torch.log(torch.exp(scales_aux) + 1)
    if scales_aux.shape[-1] == 1:
        scales_aux = scales_aux.repeat(1, 3); scales = scales.repeat(1, 3)
    return {'xyz_offsets': xyz_offsets, 'rotations': rotations, 'scales': scales, 'scales_aux': scales_aux}


def appearance_decoder(x, sd, opacity_offset=0):
    h = F.gelu(_lin(F.gelu(_lin(x, sd, "net.0")), sd, "net.2"))
    shs = _lin(h, sd, "shs").reshape(-1, 16, 3)
    opacity = torch.sigmoid(_lin(h, sd, "opacity") + opacity_offset)
    return {'shs': shs, 'opacity': opacity}
