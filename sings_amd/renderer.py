"""Host-side mirror of the reference's render glue (sings/rec/renderer/gs_renderer_single.py:12-107,
twin gs_renderer_multiple.py:12-132): same function names, argument meaning, returned keys and dtypes,
so that the reference trainer's call sites (gs_trainer.py:240-244, :561-575, :695-714) work unchanged.

Two entry points beyond the mirror:
 * ``get_render_pkgs`` concatenates several avatars before ONE raster call (gs_renderer_multiple.py:12-68);
 * ``get_render_pkg_fused`` takes CANONICAL Gaussians + joint transforms and runs the LBS-fused kernels.
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from .skinned import rasterize_skinned_gaussians


def get_render_pkg(data, human_gs_out, bg_color, scaling_modifier=1.0):
    render_pkg = render(means3D=human_gs_out['xyz'], feats=human_gs_out['shs'], opacity=human_gs_out['opacity'],
                        scales=human_gs_out['scales'], rotations=human_gs_out['rotq'], data=data,
                        scaling_modifier=scaling_modifier, bg_color=bg_color,
                        active_sh_degree=human_gs_out['active_sh_degree'])
    render_pkg['human_visibility_filter'] = render_pkg['visibility_filter']
    render_pkg['human_radii'] = render_pkg['radii']
    return render_pkg


def get_render_pkgs(data, human_gs_out_list, trans_list, rot_list, bg_color, scaling_modifier=1.0, render_mode='multi-person'):
    """Several avatars in one frame (gs_renderer_multiple.py:12-68, same signature): every avatar's ``xyz`` is translated
    IN PLACE by its entry of ``trans_list`` (:25-27; ``rot_list`` is accepted and unused, as in the reference), then the
    avatars are concatenated and rasterized once."""
    for out, trans in zip(human_gs_out_list, trans_list):
        out['xyz'] += trans[None]
    if render_mode != 'multi-person':
        raise ValueError(f'Unknown render mode: {render_mode}')
    cat = lambda k: torch.cat([h[k] for h in human_gs_out_list], dim=0).contiguous()
    render_pkg = render(means3D=cat('xyz'), feats=cat('shs'), opacity=cat('opacity'), scales=cat('scales'),
                        rotations=cat('rotq'), data=data, scaling_modifier=scaling_modifier, bg_color=bg_color,
                        active_sh_degree=human_gs_out_list[0]['active_sh_degree'])
    render_pkg['human_visibility_filter'] = render_pkg['visibility_filter']
    render_pkg['human_radii'] = render_pkg['radii']
    return render_pkg


def _settings(data, bg_color, scaling_modifier, active_sh_degree):
    return GaussianRasterizationSettings(
        image_height=int(data['image_height']), image_width=int(data['image_width']),
        tanfovx=math.tan(data['fovx'] * 0.5), tanfovy=math.tan(data['fovy'] * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=data['world_view_transform'],
        projmatrix=data['full_proj_transform'], sh_degree=active_sh_degree, campos=data['camera_center'],
        prefiltered=False, debug=False)


def render(means3D, feats, opacity, scales, rotations, data, scaling_modifier=1.0, bg_color=None, active_sh_degree=0):
    dev = means3D.device
    if bg_color is None:
        bg_color = torch.zeros(3, dtype=torch.float32, device=dev)
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True, device=dev) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    shs, rgb = (None, feats) if feats.dim() == 2 else (feats, None)
    rasterizer = GaussianRasterizer(raster_settings=_settings(data, bg_color, scaling_modifier, active_sh_degree))
    rendered_image, radii = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs, opacities=opacity,
                                       scales=scales, rotations=rotations, colors_precomp=rgb)
    raw = rendered_image
    rendered_image = torch.clamp(rendered_image, 0.0, 1.0)
    # 'render_raw' (not in the reference's dict): the unclamped image, input of the fused loss (photo_loss.py)
    return {"render": rendered_image, "render_raw": raw, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}


def get_render_pkg_fused(data, canon, A_cano2pose, bg_color, smpl_scale=None, transl=None, ext_tfs=None,
                         scaling_modifier=1.0, return_posed=False):
    """canon: dict(xyz_canon, rotmat_canon|None, scales, opacity, shs, lbs_weights, active_sh_degree)."""
    rs = _settings(data, bg_color, scaling_modifier, canon['active_sh_degree'])
    # per-call holder of the screen-space gradient, exactly the reference's idiom (gs_renderer_single.py:50-56)
    xyz = canon['xyz_canon']
    screenspace_points = None
    if torch.is_grad_enabled() and ext_tfs is None:
        screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
        screenspace_points.retain_grad()
    out = rasterize_skinned_gaussians(xyz, canon.get('rotmat_canon'), canon['scales'], canon['opacity'],
                                      canon['shs'], canon['lbs_weights'], A_cano2pose, rs, smpl_scale=smpl_scale,
                                      transl=transl, ext_tfs=ext_tfs, return_posed=return_posed,
                                      means2D=screenspace_points)
    radii = out[1]
    pkg = {"render": torch.clamp(out[0], 0.0, 1.0), "render_raw": out[0], "viewspace_points": screenspace_points,
           "visibility_filter": radii > 0,
           "radii": radii, "human_visibility_filter": radii > 0, "human_radii": radii}
    if return_posed:
        pkg.update(xyz=out[2], rotq=out[3], scales=out[4])
    return pkg


def get_render_pkgs_fused(data, canon, A_cano2pose, bg_color, smpl_scale=None, transl=None, scaling_modifier=1.0):
    """The chunk form of ``get_render_pkg_fused``: K posed frames of the same canonical Gaussians in ONE fused call (K <= 16).

    ``data``: one camera dict for all frames or a list of K dicts (same image size and field of view); ``A_cano2pose`` [K,J,4,4];
    ``transl`` [K,3] / [3] / None.  Returns the keys of ``get_render_pkg`` with a leading frame axis: 'render' [K,3,H,W],
    'radii' [K,N], 'visibility_filter' [K,N], 'viewspace_points' [K,N,3] (its ``.grad`` receives every frame's screen-space
    gradient: the densifier's statistics stay per frame)."""
    from .skinned import rasterize_skinned_frames
    datas = list(data) if isinstance(data, (list, tuple)) else None
    d0 = datas[0] if datas is not None else data
    rs = _settings(d0, bg_color, scaling_modifier, canon['active_sh_degree'])
    K = int(A_cano2pose.shape[0])
    if datas is not None:
        if len(datas) != K:
            raise ValueError(f"{len(datas)} camera dicts for {K} frames")
        rs = rs._replace(viewmatrix=torch.stack([d['world_view_transform'] for d in datas]).contiguous(),
                         projmatrix=torch.stack([d['full_proj_transform'] for d in datas]).contiguous(),
                         campos=torch.stack([d['camera_center'] for d in datas]).contiguous())
    xyz = canon['xyz_canon']
    screenspace_points = None
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros((K,) + tuple(xyz.shape), dtype=xyz.dtype, device=xyz.device, requires_grad=True) + 0
        screenspace_points.retain_grad()
    color, radii = rasterize_skinned_frames(xyz, canon.get('rotmat_canon'), canon['scales'], canon['opacity'], canon['shs'],
                                            canon['lbs_weights'], A_cano2pose, rs, smpl_scale=smpl_scale, transl=transl,
                                            means2D=screenspace_points)
    return {"render": torch.clamp(color, 0.0, 1.0), "render_raw": color, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii, "human_visibility_filter": radii > 0, "human_radii": radii}
