"""Posed-frame driver: the callers' glue either side of the fused kernels.

 * ``joint_transforms_batch``: the SMPL(-H) kinematic chain for a CHUNK of frames in J batched 4x4 products
   (reference: ``forward_chunk`` -> ``smpl_template`` -> ``batch_rigid_transform``, sings_hybrid.py:474-531,
   body_model/smpl.py:462-513) -- without the per-step template work the reference discards
   (blend shapes, the J_regressor einsum over 110k vertices, full template skinning).
 * ``amass_to_smpl_pose``: the 156 -> 72 AMASS joint selection of sings/rec/defaults/constants.py:13-18.
 * ``animate_chunk``: forward-only rendering of a chunk of posed frames through the LBS-fused kernels, one
   fused call per frame (trainer counterpart: gs_trainer.py:664-728); posed means / quaternions never exist in HBM.
   ``streams > 1`` renders that many frames at a time on separate HIP streams through pre-allocated engines
   (``FrameAnimator``): an avatar frame is a chain of short, latency-bound kernels and a few very long tile lists, so
   frames in flight together fill the GPU (150k-Gaussian avatar, 120 frames: 3 100 -> 8 550 frames/s with 4 streams).
   ``frames_per_launch = K`` (round 4): K consecutive frames of the chunk per DISPATCH through ``SkinnedFramesEngine`` -- what
   ``SinGS.forward_chunk`` + the render loop of gs_trainer.py:684-714 do for a chunk of 16 frames, in one pass over the kernels.
"""
import numpy as np
import torch

from . import _lib
from .body import SMPL_PARENTS, rodrigues
from .renderer import _settings, get_render_pkg_fused

# sings/rec/defaults/constants.py:13-18: SMPL-H (52 joints, AMASS order) -> the 24 SMPL joints
AMASS_SMPLH_TO_SMPL_JOINTS = np.arange(0, 156).reshape((-1, 3))[[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16,
                                                                   17, 18, 19, 20, 21, 22, 37]].reshape(-1)


def amass_to_smpl_pose(poses156):
    """[F,156] AMASS SMPL-H axis-angle poses -> [F,72] SMPL poses (AnimDataset_opt.py:109-110)."""
    return poses156[..., AMASS_SMPLH_TO_SMPL_JOINTS]


def joint_transforms_batch(poses, joints_rest, parents=SMPL_PARENTS):
    """poses [B, J*3] axis-angle, joints_rest [J,3] -> A [B,J,4,4].  fp32 on the GPU: ONE launch (sg_joint_transforms, one wave
    per frame); otherwise J-1 batched matmuls for the whole chunk in torch."""
    if poses.is_cuda and poses.dtype == torch.float32 and joints_rest.dtype == torch.float32 and joints_rest.shape[0] <= 64:
        from .body import joint_transforms_hip
        return joint_transforms_hip(poses, joints_rest, parents)
    B = poses.shape[0]
    J = joints_rest.shape[0]
    R = rodrigues(poses.reshape(B * J, 3)).view(B, J, 3, 3)
    rel = joints_rest.clone()
    par = torch.as_tensor(parents[1:], device=joints_rest.device, dtype=torch.long)
    rel[1:] = joints_rest[1:] - joints_rest[par]
    T = torch.zeros(B, J, 4, 4, dtype=poses.dtype, device=poses.device)
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel[None]
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.bmm(chain[parents[i]], T[:, i]))
    G = torch.stack(chain, 1)
    jh = torch.cat([joints_rest, torch.zeros_like(joints_rest[:, :1])], 1)[None, :, :, None]
    corr = torch.matmul(G, jh)
    return G - torch.cat([torch.zeros(B, J, 4, 3, dtype=G.dtype, device=G.device), corr], -1)


def forward_chunk(gs_attrs, lbs_weights, A_t2pose, inv_A_t2cano=None, transl=None, smpl_scale=None, ext_tfs=None,
                  isotropic=False, active_sh_degree=0):
    """The batched deformation of ``SinGS.forward_chunk`` (sings/rec/models/sings_hybrid.py:474-569) for callers that want
    the reference's dict of POSED Gaussians for B frames at once (gs_trainer.py:695-714 then renders them one by one);
    ``animate_chunk`` below is the fused alternative that never materialises them.

    gs_attrs: dict(xyz_canon [N,3], xyz_offsets, rot6d_canon [N,6] | None, scales [N,3], opacity [N,1], shs [N,M,3]);
    A_t2pose [B,J,4,4]: the joint transforms the reference takes from ``smpl_template(...).A`` (here:
    ``joint_transforms_batch``; shape blending needs the licensed SMPL model and stays with the caller);
    inv_A_t2cano [J,4,4] | None; transl [B,3], smpl_scale [B,1], ext_tfs = (trans [B,3], rotmat [B,3,3], scale [B,1]).
    Returns the reference's keys with the same shapes ([B,N,...], canonical tensors expanded, not copied).
    Device work: ``sg_lbs_forward`` per frame (T and vertices in one kernel, W.A on the matrix cores), the rotation kernels
    of sings_amd.rotations for 6-D -> matrix, matrix -> quaternion ([B*N] in one launch) and the quaternion product."""
    from . import rotations as R
    from .lbs import lbs_extra
    xyz_canon = gs_attrs['xyz_canon']
    dev = xyz_canon.device
    if dev.type != "cuda":
        raise RuntimeError("sings_amd.posed.forward_chunk runs on the MI355X only; there is no CPU fallback")
    B, N = int(A_t2pose.shape[0]), int(xyz_canon.shape[0])
    rot6d = gs_attrs.get('rot6d_canon')
    if not isotropic:
        rotmat_canon = R.rotation_6d_to_matrix(rot6d)
        rotq_canon = R.matrix_to_quaternion(rotmat_canon)
    else:
        rotmat_canon = torch.eye(3, device=dev).unsqueeze(0).repeat(N, 1, 1)
        rotq_canon = torch.zeros(N, 4, device=dev)
    scales = gs_attrs['scales'].unsqueeze(0).expand(B, -1, -1)
    A = A_t2pose if inv_A_t2cano is None else A_t2pose @ inv_A_t2cano.unsqueeze(0)
    xyz, _, lbs_T, _, _ = lbs_extra(A, xyz_canon.unsqueeze(0).expand(B, -1, -1), None, lbs_weights, None, disable_posedirs=True)
    if smpl_scale is not None:
        xyz = xyz * smpl_scale.unsqueeze(-1)
        scales = scales * smpl_scale.unsqueeze(-1)
    if transl is not None:
        xyz = xyz + transl.unsqueeze(1)
    rotq = R.matrix_to_quaternion(lbs_T[..., :3, :3] @ rotmat_canon.unsqueeze(0))
    if ext_tfs is not None:
        trans, rotmat, scale = ext_tfs
        xyz = trans[:, None, :] + scale[:, None] * (rotmat[:, None, ...] @ xyz[..., None]).squeeze(-1)
        scales = scale[..., None] * scales
        rotq = R.quaternion_multiply(R.matrix_to_quaternion(rotmat.contiguous())[:, None, :], rotq)
    return {
        'xyz': xyz, 'xyz_canon': xyz_canon.unsqueeze(0).expand(B, -1, -1), 'xyz_offsets': gs_attrs.get('xyz_offsets'),
        'scales': scales, 'scales_canon': scales,                    # (the reference returns the scaled tensor under both keys)
        'rotq': rotq, 'rotq_canon': rotq_canon,
        'shs': gs_attrs['shs'].unsqueeze(0).expand(B, -1, -1, -1), 'opacity': gs_attrs['opacity'].unsqueeze(0).expand(B, -1, -1),
        'active_sh_degree': active_sh_degree,
    }


class FrameAnimator:
    """Forward-only rendering of posed frames, ``streams`` frames in flight: one pre-allocated SkinnedEngine (workspaces)
    per stream, nothing allocated or synchronised per frame except one read of the pair counts per round (a frame whose
    (tile, Gaussian) pairs exceed the capacity is re-rendered after the engines have grown)."""

    def __init__(self, canon, streams=3):
        self.canon = canon
        self.dev = canon['xyz_canon'].device
        if self.dev.type != "cuda":
            raise RuntimeError("sings_amd.posed.FrameAnimator runs on the MI355X only; there is no CPU fallback")
        self.n = max(1, int(streams))
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.n)]
        self.engs, self.shape, self.cap = [], None, 0
        # SG_FLAG_LONG_ROWS (direct binning of few-tile frames: no tile list beyond 16384 entries), passed optimistically: every path
        # below reads the frames' counts before it hands an image out, and a frame that reports NUM_RENDERED_LONG_LIST is rendered
        # again without the hint -- which then stays off
        self.long_rows = True

    def _engines(self, J, W, H, cap):
        from .engine import SkinnedEngine
        c = self.canon
        P, M = int(c['xyz_canon'].shape[0]), int(c['shs'].shape[1])
        if self.shape != (P, J, W, H, M) or cap > self.cap:
            self.engs = []                                       # (drop the old workspaces before the new ones are sized)
            self.engs = [SkinnedEngine(P, J, W, H, M, self.dev, capacity_pairs=cap) for _ in range(self.n)]
            self.shape, self.cap = (P, J, W, H, M), cap
        return self.engs

    def render_round(self, jobs, bg_color, scaling_modifier=1.0):
        """jobs: up to ``streams`` tuples (camera dict, A_cano2pose [J,4,4], transl | None, smpl_scale | None,
        ext_tfs | None) -> list of clamped images [3,H,W] (the keys 'render' of get_render_pkg_fused)."""
        c = self.canon
        assert 0 < len(jobs) <= self.n
        J = int(jobs[0][1].reshape(-1, 16).shape[0])
        W, H = int(jobs[0][0]['image_width']), int(jobs[0][0]['image_height'])
        P = int(c['xyz_canon'].shape[0])
        cap = max(self.cap, 8 * P + ((W + 15) // 16) * ((H + 15) // 16), 1 << 16)
        cur = torch.cuda.current_stream(self.dev)
        while True:
            engs = self._engines(J, W, H, cap)
            outs = []
            for st in self.streams[:len(jobs)]:
                st.wait_stream(cur)
            for (cam, A, transl, smpl_scale, ext), e, st in zip(jobs, engs, self.streams):
                with torch.cuda.stream(st):
                    e.color = torch.empty((3, H, W), dtype=torch.float32, device=self.dev)    # handed to the caller
                    e.set_camera(_settings(cam, bg_color, scaling_modifier, c['active_sh_degree']), long_rows=self.long_rows)
                    e.set_frame(c['xyz_canon'], c.get('rotmat_canon'), c['lbs_weights'], A, smpl_scale, transl, ext)
                    e.forward(c['shs'], c['opacity'], c['scales'])
                    outs.append(torch.clamp(e.color, 0.0, 1.0))
                    outs[-1].record_stream(cur)
            for st in self.streams[:len(jobs)]:
                cur.wait_stream(st)
            counts = [e.num_rendered() for e in engs[:len(jobs)]]
            if self.long_rows and min(counts) == _lib.NUM_RENDERED_LONG_LIST:
                self.long_rows = False                           # a list outgrew its row: these frames again, on the plain path
                continue
            need = max(counts)
            if need <= self.cap:
                return outs
            cap = need + need // 4 + 1024


    def render_frames(self, jobs, bg_color, scaling_modifier=1.0, depth=None):
        """Pipelined variant: ``jobs`` is an iterable of the tuples ``render_round`` takes; frame i goes to stream
        i % streams and NOTHING waits between frames -- a frame is handed out (in order) once ITS forward has finished
        (its pair count comes back with an asynchronous copy, checked then), while up to ``depth`` later frames are
        already queued.  Yields (index, clamped image)."""
        from collections import deque
        c = self.canon
        depth = 2 * self.n if depth is None else max(1, int(depth))
        cur = torch.cuda.current_stream(self.dev)
        P = int(c['xyz_canon'].shape[0])
        slots = [torch.empty(2, dtype=torch.int32).pin_memory() for _ in range(depth + self.n + 1)]
        q = deque()

        def submit(i, job):
            cam, A, transl, smpl_scale, ext = job
            J = int(A.reshape(-1, 16).shape[0])
            W, H = int(cam['image_width']), int(cam['image_height'])
            cap = max(self.cap, 8 * P + ((W + 15) // 16) * ((H + 15) // 16), 1 << 16)
            if self.shape != (P, J, W, H, int(c['shs'].shape[1])) or cap > self.cap:
                torch.cuda.synchronize(self.dev)                 # (engines in flight are about to be replaced)
            e, st = self._engines(J, W, H, cap)[i % self.n], self.streams[i % self.n]
            st.wait_stream(cur)                                  # the job's tensors were produced on the caller's stream
            with torch.cuda.stream(st):
                e.color = torch.empty((3, H, W), dtype=torch.float32, device=self.dev)
                e.set_camera(_settings(cam, bg_color, scaling_modifier, c['active_sh_degree']), long_rows=self.long_rows)
                e.set_frame(c['xyz_canon'], c.get('rotmat_canon'), c['lbs_weights'], A, smpl_scale, transl, ext)
                e.forward(c['shs'], c['opacity'], c['scales'])
                img = torch.clamp(e.color, 0.0, 1.0)
                slot = slots[i % len(slots)]
                slot.copy_(e.binning[:8].view(torch.int32), non_blocking=True)      # header words 0, 1 = pair count R, flags
                ev = torch.cuda.Event()
                ev.record(st)
            img.record_stream(cur)
            q.append((i, job, img, ev, slot, e.cap))          # (the capacity THIS frame was rendered with)

        def pop():
            i, job, img, ev, slot, cap_used = q.popleft()
            ev.synchronize()
            # compared with the capacity of the engine that rendered it: an earlier pop may have grown self.cap while this
            # frame was still queued, and a count between the two capacities means a background-only image
            if int(slot[0]) > cap_used or (int(slot[1]) & 2):    # rare: grow the workspaces / drop the hint, render this frame again
                torch.cuda.synchronize(self.dev)
                if int(slot[1]) & 2:
                    self.long_rows = False
                if int(slot[0]) > cap_used:
                    need = int(slot[0])
                    self._engines(self.shape[1], self.shape[2], self.shape[3], need + need // 4 + 1024)
                img = self.render_round([job], bg_color, scaling_modifier)[0]
                torch.cuda.synchronize(self.dev)
            return i, img

        for i, job in enumerate(jobs):
            submit(i, job)
            while len(q) > depth:
                yield pop()
        while q:
            yield pop()


    # ---- K frames of the same Gaussians per dispatch (round 4: sings_amd.engine.SkinnedFramesEngine) -------------------------
    def _frame_engines(self, J, W, H, K, cap):
        from .engine import SkinnedFramesEngine
        c = self.canon
        P, M = int(c['xyz_canon'].shape[0]), int(c['shs'].shape[1])
        rot = c.get('rotmat_canon')
        key = (P, J, W, H, M, K, None if rot is None else int(rot.reshape(P, -1).shape[1]))
        if getattr(self, "_fshape", None) != key or cap > getattr(self, "_fcap", 0):
            self._fengs = []
            self._fengs = [SkinnedFramesEngine(P, J, W, H, M, K, self.dev, capacity_pairs=cap, with_rot=rot is not None,
                                               rot_width=9 if rot is None else key[6]) for _ in range(self.n)]
            self._fshape, self._fcap = key, cap
        return self._fengs

    def render_frames_batched(self, jobs, bg_color, frames_per_launch=8, scaling_modifier=1.0):
        """``render_frames`` with K = ``frames_per_launch`` consecutive frames per DISPATCH: the reference's chunk
        (``SinGS.forward_chunk``, sings_hybrid.py:474-569, rendered one frame at a time at gs_trainer.py:684-714) goes through
        every kernel of the forward once -- canonical inputs read once for the K poses, K frames' worth of workgroups for the
        latency-bound kernels.  One launch per stream in flight.  Jobs as in ``render_round`` but WITHOUT ``ext_tfs`` (the K-frame
        entry points take none) and with one ``smpl_scale`` for the batch; the cameras of a batch may differ.  Yields
        (index, clamped image) in order; images are bit-identical to the per-frame path's."""
        from collections import deque
        c = self.canon
        K = max(1, min(int(frames_per_launch), 16))
        cur = torch.cuda.current_stream(self.dev)
        P = int(c['xyz_canon'].shape[0])
        q = deque()
        state = dict(b=0)

        def launch(i0, batch, e, st):
            k = len(batch)
            cams = [j[0] for j in batch]
            A = torch.stack([j[1].reshape(-1, 16) for j in batch] + [batch[-1][1].reshape(-1, 16)] * (K - k)).contiguous()
            tr = None
            if batch[0][2] is not None:
                tr = torch.stack([j[2].reshape(3) for j in batch] + [batch[-1][2].reshape(3)] * (K - k)).contiguous()
            st.wait_stream(cur)                                  # the jobs' tensors were produced on the caller's stream
            with torch.cuda.stream(st):
                rs = _settings(cams[0], bg_color, scaling_modifier, c['active_sh_degree'])
                if any(cm is not cams[0] for cm in cams):        # per-frame cameras: stacked [K,4,4] / [K,3]
                    pad = cams + [cams[-1]] * (K - k)
                    rs = rs._replace(viewmatrix=torch.stack([cm['world_view_transform'] for cm in pad]).contiguous(),
                                     projmatrix=torch.stack([cm['full_proj_transform'] for cm in pad]).contiguous(),
                                     campos=torch.stack([cm['camera_center'] for cm in pad]).contiguous())
                e.set_camera(rs, long_rows=self.long_rows)
                e.set_frames(c['xyz_canon'], c.get('rotmat_canon'), c['lbs_weights'], A, batch[0][3], tr)
                e.forward(c['shs'], c['opacity'], c['scales'])
                img = torch.clamp(e.color, 0.0, 1.0)             # [K,3,H,W]: a new tensor, the engine's image buffer is reused
                ev = torch.cuda.Event(); ev.record(st)
            img.record_stream(cur)
            return img, ev

        def submit(i0, batch):
            for j in batch:
                if j[4] is not None:
                    raise ValueError("render_frames_batched: ext_tfs are not supported by the K-frame entry points "
                                     "(use render_frames)")
                if j[3] is not batch[0][3] and not (j[3] is not None and batch[0][3] is not None and torch.equal(j[3], batch[0][3])):
                    raise ValueError("render_frames_batched: one smpl_scale per batch of frames")
            J = int(batch[0][1].reshape(-1, 16).shape[0])
            W, H = int(batch[0][0]['image_width']), int(batch[0][0]['image_height'])
            cap = max(getattr(self, "_fcap", 0), 8 * P + ((W + 15) // 16) * ((H + 15) // 16), 1 << 16)
            engs = self._frame_engines(J, W, H, K, cap)
            e, st = engs[state['b'] % self.n], self.streams[state['b'] % self.n]
            state['b'] += 1
            img, ev = launch(i0, batch, e, st)
            q.append((i0, batch, img, ev, e, st))

        def pop():
            i0, batch, img, ev, e, st = q.popleft()
            ev.synchronize()
            with torch.cuda.stream(st):
                Rs = e.num_rendered()                            # (this engine has nothing else queued: one launch per engine)
            long_list = self.long_rows and min(Rs[:len(batch)]) == _lib.NUM_RENDERED_LONG_LIST
            if max(Rs[:len(batch)]) > e.cap or long_list:        # rare: grow the workspaces / drop the hint, render this batch again
                torch.cuda.synchronize(self.dev)
                if long_list:
                    self.long_rows = False
                # (a NEW set of engines either way: the queued batches keep theirs, whose counts are still to be read)
                need = max(max(Rs[:len(batch)]), self._fcap - self._fcap // 5)
                J, W, H = self._fshape[1], self._fshape[2], self._fshape[3]
                e2 = self._frame_engines(J, W, H, K, need + need // 4 + 1024)[0]
                img, ev = launch(i0, batch, e2, self.streams[0])
                torch.cuda.synchronize(self.dev)
                Rs2 = e2.num_rendered()
                if max(Rs2[:len(batch)]) > e2.cap:               # (both at once: the plain path has now counted the batch)
                    need = max(Rs2[:len(batch)])
                    e2 = self._frame_engines(J, W, H, K, need + need // 4 + 1024)[0]
                    img, ev = launch(i0, batch, e2, self.streams[0])
                    torch.cuda.synchronize(self.dev)
            for k in range(len(batch)):
                yield i0 + k, img[k]

        batch, i0, i = [], 0, 0
        for i, job in enumerate(jobs):
            if not batch:
                i0 = i
            batch.append(job)
            if len(batch) == K:
                while len(q) >= self.n:
                    yield from pop()
                submit(i0, batch)
                batch = []
        if batch:
            while len(q) >= self.n:
                yield from pop()
            submit(i0, batch)
        while q:
            yield from pop()


@torch.no_grad()
def animate_chunk(canon, poses, joints_rest, A_t2cano, cameras, bg_color, transl=None, smpl_scale=None, ext_tfs=None,
                  parents=SMPL_PARENTS, chunk_size=16, streams=1, animator=None, frames_per_launch=1):
    """Renders frames ``poses[f]`` with camera dict ``cameras[f]`` (or one shared dict); yields (f, image[3,H,W]).

    canon: dict(xyz_canon, rotmat_canon|None, scales, opacity, shs, lbs_weights, active_sh_degree);
    transl [F,3] or None; ext_tfs: per-frame tuple (trans[F,3], rotmat[F,3,3], scale[F,1]) or None.
    streams > 1 (or an existing ``animator``): that many frames in flight (FrameAnimator); same images.
    frames_per_launch = K > 1 (without ext_tfs): K consecutive frames per DISPATCH (FrameAnimator.render_frames_batched), one such
    launch per stream in flight; same images."""
    F = poses.shape[0]
    inv_cano = torch.inverse(A_t2cano)
    batched = int(frames_per_launch) > 1 and ext_tfs is None
    if animator is None and (streams > 1 or batched):
        animator = FrameAnimator(canon, streams)
    if animator is not None:
        # ONE pipelined pass over all chunks: the joint transforms of chunk k + 1 are computed (on the caller's stream) while
        # the frames of chunk k are still in flight -- nothing drains between chunks
        # (the kinematic chain is ~80 small launches per chunk whatever its size: at least 128 frames per chain here)
        jt = max(int(chunk_size), 128)

        def jobs():
            for c0 in range(0, F, jt):
                A = joint_transforms_batch(poses[c0:c0 + jt], joints_rest, parents) @ inv_cano[None]
                for i in range(A.shape[0]):
                    f = c0 + i
                    yield (cameras[f] if isinstance(cameras, (list, tuple)) else cameras, A[i],
                           None if transl is None else transl[f], smpl_scale,
                           None if ext_tfs is None else (ext_tfs[0][f], ext_tfs[1][f], ext_tfs[2][f]))
        if batched:
            yield from animator.render_frames_batched(jobs(), bg_color, frames_per_launch=int(frames_per_launch))
        else:
            yield from animator.render_frames(jobs(), bg_color)
        return
    for c0 in range(0, F, chunk_size):
        A = joint_transforms_batch(poses[c0:c0 + chunk_size], joints_rest, parents) @ inv_cano[None]
        for i in range(A.shape[0]):
            f = c0 + i
            cam = cameras[f] if isinstance(cameras, (list, tuple)) else cameras
            ext = None if ext_tfs is None else (ext_tfs[0][f], ext_tfs[1][f], ext_tfs[2][f])
            pkg = get_render_pkg_fused(cam, canon, A[i], bg_color, smpl_scale=smpl_scale,
                                       transl=None if transl is None else transl[f], ext_tfs=ext)
            yield f, pkg["render"]
