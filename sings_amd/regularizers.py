"""Geometry-preserving regularisers of the SinGS trainer through the C ABI.

Same class / function names, constructor arguments and forward signatures as the reference
(sings/rec/losses/loss_items.py; pytorch3d.loss.mesh_edge_loss for the mesh term), so
``gs_trainer.py:168-194, 355-399`` can use them unchanged:

    L2Norm(**lambdas)(human_gs_out)                                   loss_items.py:15-54
    GaussiansEdgeLoss(K=9)(human_gs_out)                              :57-90
    RegionLaplacianLoss_v2(verts, edges, vertex_labels, region_weights=...)(x), .forward_hands(x)   :93-192
    mesh_edge_loss(verts, edges)                                      pytorch3d.loss.mesh_edge_loss(mesh, 0)

Each returns a 0-dim loss tensor with autograd; the gradient is computed in the same kernel pass as the value and only
scaled in backward.  Static graph structure (CSR of same-label edges, degrees, per-vertex region scale) is built once on
the host with numpy, like ``reset_laplacians`` does with torch.  No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .rasterizer import _ptr


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"sings_amd.regularizers.{what}: tensors must live on the GPU (no CPU fallback)")


def _csr(num_verts, edges):
    """edges [E,2] unique undirected -> (row_ptr int32 [V+1], col int32 [2E]) with both directions."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    src = np.concatenate([e[:, 0], e[:, 1]]); dst = np.concatenate([e[:, 1], e[:, 0]])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    row_ptr = np.zeros(num_verts + 1, np.int64)
    np.add.at(row_ptr, src + 1, 1)
    return np.cumsum(row_ptr).astype(np.int32), dst.astype(np.int32)


class _ScaledGrad(torch.autograd.Function):
    """loss (0-dim) whose gradients w.r.t. the listed inputs were already computed by the kernel."""

    @staticmethod
    def forward(ctx, loss, n_inputs, *tensors):
        inputs, grads = tensors[:n_inputs], tensors[n_inputs:]
        ctx.save_for_backward(*grads)
        ctx.n = n_inputs
        return loss.view_as(loss)          # (a view: no copy kernel; `loss` is the kernel's own output buffer)

    @staticmethod
    def backward(ctx, g):
        # ONE multi-tensor launch for all the pre-computed gradients instead of one multiplication kernel per tensor
        ds = list(ctx.saved_tensors)
        return (None, None) + tuple(torch._foreach_mul(ds, g.reshape(()))) + (None,) * ctx.n


def _attach(loss, inputs, grads, filled=True):
    """``filled=False``: the kernel that writes ``loss`` has NOT run yet (GaussiansEdgeLoss.prepare): nothing may copy it now --
    with no input that requires grad (evaluation under no_grad, detached scales) the caller gets a VIEW of the buffer the later
    kernel fills, never a clone of uninitialised memory (round 3 cloned here: garbage in validation)."""
    keep_i, keep_g = [], []
    for t, d in zip(inputs, grads):
        if t is not None and d is not None and t.requires_grad and torch.is_grad_enabled():
            keep_i.append(t); keep_g.append(d)
    if not keep_i:
        return loss.clone() if filled else loss.detach()
    return _ScaledGrad.apply(loss, len(keep_i), *keep_i, *keep_g)


class L2Norm(torch.nn.Module):
    def __init__(self, lambda_xyz_offsets=0.005, lambda_scales_diff=0.005, lambda_max_scale=0.001,
                 max_scale_threshold=0.008, lambda_min_opacity=0.0001, min_opacity_threshold=0.2):
        super().__init__()
        self._l = (lambda_xyz_offsets, lambda_scales_diff, lambda_max_scale, max_scale_threshold, lambda_min_opacity,
                   min_opacity_threshold)
        self._lam = {}                 # device -> the six scalars on that device (uploaded once, not per call)

    def forward(self, human_gs_out):
        lib = _lib.load()
        off = human_gs_out['xyz_offsets'].contiguous().float()
        sc = human_gs_out['scales'].contiguous().float()
        op = human_gs_out['opacity'].contiguous().float() if 'opacity' in human_gs_out else None
        _need_gpu(off, "L2Norm")
        dev, N = off.device, int(off.shape[0])
        lam = self._lam.get(dev)
        if lam is None:
            lam = self._lam[dev] = torch.tensor(self._l, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.sg_reg_ws_bytes(N)), dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        d_off, d_sc = torch.empty_like(off), torch.empty_like(sc)
        d_op = torch.empty_like(op) if op is not None else None
        with torch.cuda.device(dev):
            _lib.check(lib.sg_l2norm_reg(N, _ptr(off), _ptr(sc), _ptr(op), _ptr(lam), _ptr(ws), _ptr(loss), None,
                                         _ptr(d_off), _ptr(d_sc), _ptr(d_op), _stream(dev)), "l2norm")
        # (what the kernel computed besides the value: d loss / d input for an upstream gradient of 1 -- a caller that stages its
        #  backward pass by hand, sings_amd.train_step.AvatarStep, reads them here instead of going through autograd)
        self.last_grads = {"xyz_offsets": d_off, "scales": d_sc, "opacity": d_op}
        return _attach(loss[0], [human_gs_out['xyz_offsets'], human_gs_out['scales'], human_gs_out.get('opacity')],
                       [d_off, d_sc, d_op])


class GaussiansEdgeLoss(torch.nn.Module):
    def __init__(self, K=9, eps=1e-12):
        super().__init__()
        self._K, self._eps = K, eps

    def prepare(self, human_gs_out):
        """First half on the current stream: the two search grids over the points (small, latency-bound launches), the output
        buffers and the autograd node.  Returns the loss tensor -- its VALUE (and the gradient the node hands back) arrives
        with `finish()`, which runs the neighbour query and the loss on the stream that is current THEN (the caller orders
        the two streams).  A trainer builds the grids as soon as the positions exist and holds the query -- one kernel that
        fills the GPU -- until it can run beside kernels with idle issue slots; creating the node early keeps it BEHIND the
        rasterizer's node in autograd's execution order, so that the backward composite is not held up by the hand-over of
        this gradient (`sings_amd.train_step.AvatarStep`)."""
        lib = _lib.load()
        if getattr(self, "_pending", None) is not None:
            # ONE pending query per module: a second prepare() would orphan the first node -- its loss and gradient buffers
            # would never be written and autograd would hand on whatever they held
            raise RuntimeError("GaussiansEdgeLoss.prepare() called twice without finish(): finish() (or forward()) the first one")
        verts = human_gs_out['xyz_canon'].detach().contiguous().float()      # edge lengths are detached (:75)
        sc = human_gs_out['scales'].contiguous().float()
        _need_gpu(verts, "GaussiansEdgeLoss")
        dev, N = verts.device, int(verts.shape[0])
        ws = torch.empty(int(lib.sg_knn_ws_bytes(N)), dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        d_sc = torch.empty_like(sc)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_gaussian_edge_prepare(N, _ptr(verts), _ptr(ws), _stream(dev)), "gaussian edge loss (grids)")
        self._pending = (N, verts, sc, ws, loss, d_sc, torch.cuda.current_stream(dev))
        self.last_grads = {"scales": d_sc}                       # (filled by finish(), like the loss; see L2Norm.last_grads)
        return _attach(loss[0], [human_gs_out['scales']], [d_sc], filled=False)

    def finish(self):
        lib = _lib.load()
        if getattr(self, "_pending", None) is None:
            raise RuntimeError("GaussiansEdgeLoss.finish() without a prepare() before it")
        N, verts, sc, ws, loss, d_sc, st0 = self._pending
        dev = sc.device                              # (same stream as prepare(), or one the caller has ordered behind it AND
        #                                               told the allocator about: the buffers were allocated there)
        if torch.cuda.is_current_stream_capturing() and torch.cuda.current_stream(dev) != st0:
            # a SECOND side stream inside a HIP-graph capture: the buffers of prepare() would need record_stream across streams
            # of the capture, and hipStreamEndCapture crashed on exactly that (round 3) -- refuse instead of dying later
            self._pending = None                     # (the query is dropped: the module stays usable after the error)
            raise RuntimeError("GaussiansEdgeLoss.finish(): inside a HIP-graph capture the query must run on the stream that ran "
                               "prepare() (keep the whole regulariser on ONE side stream)")
        self._pending = None
        with torch.cuda.device(dev):
            _lib.check(lib.sg_gaussian_edge_finish(N, self._K, _ptr(sc), _ptr(ws), None, _ptr(loss), None, _ptr(d_sc),
                                                   _stream(dev)), "gaussian edge loss (query)")

    def abort(self):
        """Drop a prepared query that will never be finished (the step between prepare() and finish() raised -- an overflow with
        on_overflow='raise', an allocation failure, bad input): the loss tensor prepare() returned must not be used; the module is
        ready for the next prepare().  No-op when nothing is pending."""
        self._pending = None

    def forward(self, human_gs_out):
        out = self.prepare(human_gs_out)
        try:
            self.finish()
        except BaseException:
            self.abort()
            raise
        return out


def knn_mean_edge(xyz, K=9):
    """[N] mean distance to the K-1 nearest other points (exact)."""
    lib = _lib.load()
    xyz = xyz.detach().contiguous().float()
    _need_gpu(xyz, "knn_mean_edge")
    dev, N = xyz.device, int(xyz.shape[0])
    ws = torch.empty(int(lib.sg_knn_ws_bytes(N)), dtype=torch.uint8, device=dev)
    out = torch.empty(N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.sg_gaussian_edge_loss(N, K, _ptr(xyz), None, _ptr(ws), _ptr(out), None, None, None, _stream(dev)),
                   "knn")
    return out


class RegionLaplacianLoss_v2(torch.nn.Module):
    def __init__(self, verts, edges, vertex_labels, faces=None, region_weights=None, laplacian_type="standard"):
        """only for unique edges.  ``region_weights``: sequence indexed by label (what the reference's ``parse_weights``
        returns) or a dict label -> weight."""
        super().__init__()
        if laplacian_type not in ("standard", "cotangent"):
            raise NotImplementedError("laplacian_type 'norm' raises in the reference too (loss_items.py:110-112)")
        self.laplacian_type = laplacian_type
        self.dev = verts.device
        self.reset_laplacians(verts, edges, vertex_labels, faces)
        n_lab = int(self.unique_labels.max()) + 1
        if region_weights is None:
            w = np.ones(n_lab)
        elif isinstance(region_weights, dict):
            w = np.ones(n_lab)
            for k, v in region_weights.items():
                w[int(k)] = v
        else:
            w = np.asarray(region_weights, dtype=np.float64)
        self.weights = w

    def reset_laplacians(self, verts, edges, vertex_labels, faces=None):
        lab = vertex_labels.detach().cpu().numpy() if torch.is_tensor(vertex_labels) else np.asarray(vertex_labels)
        lab = lab.astype(np.int64)
        if (lab < 0).any():
            raise ValueError("vertex labels must be >= 0")
        if self.laplacian_type == "cotangent":
            if faces is None:
                raise ValueError("faces is supposed to be provided when using cotangent laplacian.")      # loss_items.py:130-131
            self._reset_cot(verts, lab, faces)
            return
        e = edges.detach().cpu().numpy() if torch.is_tensor(edges) else np.asarray(edges)
        e = e.astype(np.int64).reshape(-1, 2)
        V = int(lab.shape[0])
        same = lab[e[:, 0]] == lab[e[:, 1]]                     # loss_items.py:139: edges whose endpoints share the label
        row_ptr, col = _csr(V, e[same])
        deg = np.diff(row_ptr).astype(np.float64)
        self.V = V
        self.labels = lab
        self.unique_labels = np.unique(lab)
        self.counts = np.bincount(lab, minlength=int(lab.max()) + 1)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(self.dev)
        self.row_ptr, self.col = t(row_ptr, np.int32), t(col, np.int32)
        self.deg_inv = t(np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0), np.float32)
        self._vscale_cache = {}

    def _reset_cot(self, verts, lab, faces, eps=1e-12):
        """loss_items.py:150-165: per label, the faces with ANY vertex of that label, their vertices (the partitions overlap), the
        cotangent weights of pytorch3d.ops.cot_laplacian on those faces (fp32, its formula order; duplicates of an edge summed).
        Stored as ONE stack of weighted rows over the global vertex array + its transpose (sg_rows_laplacian)."""
        v = (verts.detach().cpu().numpy() if torch.is_tensor(verts) else np.asarray(verts)).astype(np.float32)
        f = (faces.detach().cpu().numpy() if torch.is_tensor(faces) else np.asarray(faces)).astype(np.int64).reshape(-1, 3)
        V = int(lab.shape[0])
        self.V, self.labels = V, lab
        self.unique_labels = np.unique(lab)
        n_lab = int(lab.max()) + 1
        fl = lab[f]
        rows_g, cols_g, vals, row_label, row_vertex = [], [], [], [], []
        self.counts = np.zeros(n_lab, np.int64)                 # rows (= vertices of the partition) per label
        base = 0
        for label in self.unique_labels:
            sel = f[(fl == label).any(axis=1)]
            inc = np.unique(sel)
            local = np.searchsorted(inc, sel)
            p0, p1, p2 = v[sel[:, 0]], v[sel[:, 1]], v[sel[:, 2]]
            A = np.linalg.norm(p1 - p2, axis=1); B = np.linalg.norm(p0 - p2, axis=1); Cc = np.linalg.norm(p0 - p1, axis=1)
            s_ = np.float32(0.5) * (A + B + Cc)
            area = np.sqrt(np.maximum(s_ * (s_ - A) * (s_ - B) * (s_ - Cc), np.float32(eps)))
            A2, B2, C2 = A * A, B * B, Cc * Cc
            cot = np.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], axis=1) / np.float32(4.0)
            ii = local[:, [1, 2, 0]].reshape(-1); jj = local[:, [2, 0, 1]].reshape(-1); w = cot.reshape(-1)
            r = np.concatenate([ii, jj]); c = np.concatenate([jj, ii]); w2 = np.concatenate([w, w])          # L += L^T
            rows_g.append(base + r); cols_g.append(inc[c]); vals.append(w2)
            row_label.append(np.full(len(inc), label)); row_vertex.append(inc)
            self.counts[label] = len(inc)
            base += len(inc)
        R = base
        r = np.concatenate(rows_g); c = np.concatenate(cols_g); w = np.concatenate(vals).astype(np.float64)
        key = r * V + c                                          # coalesce duplicate (row, column) entries
        uk, inv = np.unique(key, return_inverse=True)
        ws = np.bincount(inv, weights=w, minlength=len(uk))
        r, c = uk // V, uk % V                                   # sorted by row, then column: CSR order
        row_ptr = np.zeros(R + 1, np.int64); np.add.at(row_ptr, r + 1, 1); row_ptr = np.cumsum(row_ptr)
        order = np.lexsort((r, c))                               # transposed: by column, then row
        t_row_ptr = np.zeros(V + 1, np.int64); np.add.at(t_row_ptr, c + 1, 1); t_row_ptr = np.cumsum(t_row_ptr)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(self.dev)
        self.R = R
        self.row_label = np.concatenate(row_label)
        self.row_ptr, self.col, self.val = t(row_ptr, np.int32), t(c, np.int32), t(ws, np.float32)
        self.t_row_ptr, self.t_row, self.t_val = t(t_row_ptr, np.int32), t(r[order], np.int32), t(ws[order], np.float32)
        self._vscale_cache = {}

    def _rscale(self, C, per_label):
        key = ("rows", C, tuple(np.round(per_label, 12)))
        if key not in self._vscale_cache:
            rs = per_label[self.row_label] / (self.counts[self.row_label] * float(C))
            self._vscale_cache[key] = torch.from_numpy(rs.astype(np.float32)).to(self.dev)
        return self._vscale_cache[key]

    def _run_cot(self, x, per_label):
        lib = _lib.load()
        xin = x
        x = x.contiguous().float()
        _need_gpu(x, "RegionLaplacianLoss_v2")
        if x.shape[0] != self.V:
            raise ValueError(f"expected {self.V} rows, got {x.shape[0]}")
        C_ = int(x.numel() // self.V)
        rs = self._rscale(C_, per_label)
        dev = x.device
        ws = torch.empty(int(lib.sg_reg_ws_bytes(self.R)), dtype=torch.uint8, device=dev)
        g = torch.empty((self.R, C_), dtype=torch.float32, device=dev); dx = torch.empty_like(x)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_rows_laplacian(self.R, self.V, C_, _ptr(x), _ptr(self.row_ptr), _ptr(self.col), _ptr(self.val),
                                             _ptr(rs), _ptr(self.t_row_ptr), _ptr(self.t_row), _ptr(self.t_val), _ptr(ws), _ptr(g),
                                             _ptr(loss), None, _ptr(dx), _stream(dev)), "region laplacian (cotangent)")
        return _attach(loss[0], [xin], [dx.view_as(xin)])

    def _vscale(self, C, per_label):
        key = (C, tuple(np.round(per_label, 12)))
        if key not in self._vscale_cache:
            vs = per_label[self.labels] / (self.counts[self.labels] * float(C))
            self._vscale_cache[key] = torch.from_numpy(vs.astype(np.float32)).to(self.dev)
        return self._vscale_cache[key]

    def _run(self, x, per_label):
        if self.laplacian_type == "cotangent":
            return self._run_cot(x, per_label)
        lib = _lib.load()
        xin = x
        x = x.contiguous().float()
        _need_gpu(x, "RegionLaplacianLoss_v2")
        if x.shape[0] != self.V:
            raise ValueError(f"expected {self.V} rows, got {x.shape[0]}")
        C_ = int(x.numel() // self.V)
        vs = self._vscale(C_, per_label)
        dev = x.device
        ws = torch.empty(int(lib.sg_reg_ws_bytes(self.V)), dtype=torch.uint8, device=dev)
        g = torch.empty_like(x); dx = torch.empty_like(x)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sg_region_laplacian(self.V, C_, _ptr(x), _ptr(self.row_ptr), _ptr(self.col), _ptr(self.deg_inv),
                                               _ptr(vs), _ptr(ws), _ptr(g), _ptr(loss), None, _ptr(dx), _stream(dev)),
                       "region laplacian")
        return _attach(loss[0], [xin], [dx.view_as(xin)])

    def forward(self, x):
        per = np.zeros(len(self.counts))
        per[self.unique_labels] = self.weights[self.unique_labels]
        return self._run(x, per)

    def forward_hands(self, x, hand_strength=1000):
        per = np.zeros(len(self.counts))
        per[[6, 7]] = hand_strength                              # loss_items.py:172-180
        return self._run(x[:self.V] if x.shape[0] != self.V else x, per)


class _MeshEdges:
    """CSR of a mesh's unique edges on the device (``MeshEdges`` for callers that build it once per topology)."""

    def __init__(self, num_verts, edges, dev):
        row_ptr, col = _csr(num_verts, edges)
        self.V, self.E = int(num_verts), int(np.asarray(edges).reshape(-1, 2).shape[0])
        self.row_ptr = torch.from_numpy(row_ptr).to(dev); self.col = torch.from_numpy(col).to(dev)


MeshEdges = _MeshEdges


def mesh_edge_loss(verts, edges, _cache={}):
    """pytorch3d.loss.mesh_edge_loss(Meshes([verts], [faces]), target_length=0.0) for ONE mesh whose unique edges are
    ``edges`` [E,2] (``mesh.edges_packed()``): mean_e |v0 - v1|^2."""
    lib = _lib.load()
    x = verts.contiguous().float()
    _need_gpu(x, "mesh_edge_loss")
    dev = x.device
    if isinstance(edges, _MeshEdges):                   # built once by the caller: MeshEdges(num_verts, edges, device)
        m = edges
    else:
        # look the CSR up FIRST; the edge list is converted (a D2H copy + host synchronisation for a device tensor) only on a
        # miss.  Key: storage address + version counter + shape of a tensor, so that a recycled Python id or an in-place
        # edit cannot hit a stale entry; the cached entry keeps the tensor alive, so its address cannot be reused meanwhile.
        if torch.is_tensor(edges):
            key = ("t", edges.data_ptr(), edges._version, tuple(edges.shape), str(edges.device), int(x.shape[0]), str(dev))
        else:
            e_np = np.ascontiguousarray(np.asarray(edges))
            key = ("n", e_np.shape, hash(e_np.tobytes()), int(x.shape[0]), str(dev))
        hit = _cache.get(key)
        if hit is None:
            e_np = edges.detach().cpu().numpy() if torch.is_tensor(edges) else e_np
            _cache.clear()
            hit = _cache[key] = (_MeshEdges(int(x.shape[0]), e_np, dev), edges)
        m = hit[0]
    ws = torch.empty(int(lib.sg_reg_ws_bytes(m.V)), dtype=torch.uint8, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev); dx = torch.empty_like(x)
    with torch.cuda.device(dev):
        _lib.check(lib.sg_mesh_edge_loss(m.V, m.E, _ptr(x), _ptr(m.row_ptr), _ptr(m.col), _ptr(ws), _ptr(loss), None,
                                         _ptr(dx), _stream(dev)), "mesh edge loss")
    return _attach(loss[0], [verts], [dx])
