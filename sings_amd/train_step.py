"""One SinGS training step on one posed frame, composed from the library's pieces (what ``scripts/train_avatar.py`` ->
``GaussianTrainer.train`` does per iteration, SURVEY.md 3.1, without the optimiser / densification bookkeeping):

    decode_attributes (tri-plane + decoders)          sings_hybrid.py:249-313      sings_amd/decode.py
    fused LBS + projection + binning + composite      sings_hybrid.py:398-428, gs_renderer_single.py:12-107   sings_amd/skinned.py
    clamp + L1 + SSIM                                 losses/loss.py:55-69         sings_amd/photo_loss.py
    regularisers (optional)                           gs_trainer.py:355-399        sings_amd/regularizers.py
    backward through all of it

Everything between the parameters and the scalar loss runs in HIP kernels (library GEMMs for the decoders' forward /
input-gradient products); torch autograd only strings the stages together.  The regularisers depend on the decoded
attributes only, not on the render: with ``overlap_regularisers`` (default) they run on a second HIP stream next to the
LBS + raster + photometric-loss chain -- the k-NN query (a chain of dependent look-ups per lane) and the render kernels
(tile-imbalance tails) fill each other's holes; the streams fork after the decode and join at the loss sum, also inside a
captured HIP graph.
"""
import torch

from .decode import decode_attributes
from .photo_loss import photometric_loss
from .skinned import rasterize_skinned_gaussians


class AvatarStep(torch.nn.Module):
    def __init__(self, xyz_anchor, lbs_weights, triplane, geometry_dec, appearance_dec, l1_w=0.8, ssim_w=0.2,
                 thickness_factor=1.0, scaling_multiplier=None, l2_norm=None, gaussian_connect=None, gaussian_connect_w=0.0,
                 overlap_regularisers=True):
        super().__init__()
        self.overlap_regularisers, self._side = bool(overlap_regularisers), None
        self.xyz = torch.nn.Parameter(xyz_anchor.detach().clone())
        self.register_buffer("lbs_weights", lbs_weights.detach().clone())
        self.triplane, self.geometry_dec, self.appearance_dec = triplane, geometry_dec, appearance_dec
        self.l1_w, self.ssim_w, self.thickness_factor = l1_w, ssim_w, thickness_factor
        self.scaling_multiplier = scaling_multiplier
        self.l2_norm, self.gaussian_connect, self.gaussian_connect_w = l2_norm, gaussian_connect, gaussian_connect_w

    def forward(self, A_cano2pose, raster_settings, gt_rgb, mask, bg_color, smpl_scale=None, transl=None):
        attrs = decode_attributes(self.xyz, self.triplane, self.geometry_dec, self.appearance_dec, self.thickness_factor,
                                  self.scaling_multiplier)
        # anisotropic: the decoder's 6-D rotations go to the fused kernels as they are (rotation_6d_to_matrix of
        # sings_hybrid.py:356-357 runs inside them); isotropic: None
        rot = attrs["rot6d_canon"]
        reg = {}

        def regularisers():
            if self.l2_norm is not None:
                reg["l2"] = self.l2_norm({"xyz_offsets": attrs["xyz_offsets"], "scales": attrs["scales"],
                                          "opacity": attrs["opacity"]})
            if self.gaussian_connect is not None and self.gaussian_connect_w > 0:
                reg["gaussian_connect_loss"] = self.gaussian_connect_w * self.gaussian_connect(
                    {"xyz_canon": attrs["xyz_canon"], "scales": attrs["scales"]})

        has_reg = self.l2_norm is not None or (self.gaussian_connect is not None and self.gaussian_connect_w > 0)
        side = None
        if has_reg and self.overlap_regularisers and attrs["xyz_canon"].is_cuda:
            dev = attrs["xyz_canon"].device
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(dev)
            side, cur = self._side, torch.cuda.current_stream(dev)
            side.wait_stream(cur)                                # fork: the decoded attributes are complete
            with torch.cuda.stream(side):
                regularisers()
        color, radii = rasterize_skinned_gaussians(attrs["xyz_canon"], rot, attrs["scales"], attrs["opacity"], attrs["shs"],
                                                   self.lbs_weights, A_cano2pose, raster_settings, smpl_scale=smpl_scale,
                                                   transl=transl)
        loss_dict, extras = photometric_loss(color, gt_rgb, mask, bg_color, self.l1_w, self.ssim_w)
        if side is not None:
            cur.wait_stream(side)                                # join before the loss terms meet
            for v in reg.values():
                v.record_stream(cur)
        elif has_reg:
            regularisers()
        loss_dict.update(reg)
        loss = torch.stack([v.reshape(()) for v in loss_dict.values()]).sum()      # two launches, not one addition per term
        loss_dict["loss"] = loss
        return loss, loss_dict, {"render_raw": color, "radii": radii, "attrs": attrs, **extras}
