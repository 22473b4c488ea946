"""TEST INFRASTRUCTURE ONLY -- CPU (torch fp32, autograd) restatement of the reference's regularisers.

* ``L2Norm`` follows sings/rec/losses/loss_items.py:15-54 line by line (pure torch in the reference: pinned by the
  golden vectors of tests/golden/gen_reg_golden.py, which runs the reference class itself).
* ``gaussians_edge_loss`` follows :57-90; ``region_laplacian_loss`` / ``forward_hands`` follow :93-192.  Both sit on
  pytorch3d (``knn_points``, ``laplacian``), pulled un-pinned from its main branch by install_all.sh:21 and absent
  from this image: PARITY UNPINNED for those two primitives, restated from their published definitions --
    knn_points(p, p, K): the K nearest points by squared Euclidean distance, ascending (self first, distance 0);
    laplacian(verts, edges): L = D^-1 A - I, A the symmetric 0/1 adjacency of the edge list, D^-1 = 0 where deg = 0;
    mesh_edge_loss(mesh, target_length=0): mean over the mesh's unique edges of |v0 - v1|^2 (one mesh).
  The golden generator runs the REFERENCE classes on top of these restated primitives, so everything except the two
  primitives themselves is the reference's own arithmetic.
* ``cot_laplacian`` (round 5) restates pytorch3d.ops.cot_laplacian (ops/laplacian_matrices.py of the 0.7 series; also absent
  here: PARITY UNPINNED): per face, side lengths A, B, C opposite v0, v1, v2, Heron's area clamped at eps before the root,
  cot(angle at v_k) = (sum of the other two squared sides - the opposite one) / (4 area); L[v1,v2] = cot a, L[v2,v0] = cot b,
  L[v0,v1] = cot c, then L += L^T -- the off-diagonal weights only, NO diagonal; second result: 1 / (sum of the areas of the
  faces around a vertex).  ``region_laplacian_cot_loss`` follows loss_items.py:150-165 (regions = faces with ANY vertex of the
  label: the partitions overlap) and :183-192.  It is checked against hand-computed cotangents (tests/test_oracle_reg.py).
Only tests/ may import this module.
"""
import torch


def knn_points(p1, p2, K):
    """minimal stand-in for pytorch3d.ops.knn_points on [1,N,3] inputs: (dists [1,N,K] squared, idx [1,N,K], None)."""
    d = torch.cdist(p1[0].double(), p2[0].double()) ** 2
    dists, idx = torch.topk(d, K, dim=1, largest=False, sorted=True)
    return dists.float()[None], idx[None], None


def laplacian(verts, edges):
    V = verts.shape[0]
    e0, e1 = edges.unbind(1)
    A = torch.zeros((V, V), dtype=torch.float32)
    A[e0, e1] = 1.0; A[e1, e0] = 1.0
    deg = A.sum(1)
    deg_inv = torch.where(deg > 0, 1.0 / deg, deg)
    return deg_inv[:, None] * A - torch.eye(V)


def l2norm(human_gs_out, lambda_xyz_offsets=0.005, lambda_scales_diff=0.005, lambda_max_scale=0.001,
           max_scale_threshold=0.008, lambda_min_opacity=0.0001, min_opacity_threshold=0.2):
    xyz_offsets = human_gs_out['xyz_offsets']
    scales = human_gs_out['scales'][:, 0]
    scales_diff = scales - scales.mean(dim=0)
    idx = scales > max_scale_threshold
    loss = lambda_xyz_offsets * xyz_offsets.norm() + lambda_scales_diff * scales_diff.norm() + \
        lambda_max_scale * scales[idx].norm()
    if 'opacity' in human_gs_out:
        opacity = human_gs_out['opacity']
        oi = opacity < min_opacity_threshold
        loss = loss + lambda_min_opacity * (0.5 - opacity[oi]).norm()
    return loss


def gaussians_edge_loss(human_gs_out, K=9):
    verts = human_gs_out['xyz_canon']
    scales = human_gs_out['scales'][:, 0]
    _, idx_knn, _ = knn_points(verts.unsqueeze(0), verts.unsqueeze(0), K=K)
    edge_vectors = verts[idx_knn[0, :, 1:]] - verts.unsqueeze(1)
    edge_lengths = torch.norm(edge_vectors, dim=-1).mean(dim=-1, keepdim=True).detach()
    return ((scales.unsqueeze(1) - edge_lengths) ** 2).mean(), edge_lengths[:, 0]


def _region_parts(verts, edges, labels):
    parts = []
    edge_label = labels[edges]
    for label in torch.unique(labels):
        inc = labels == label
        sel = edges[torch.all(edge_label == label, dim=1)]
        uniq, inv = torch.unique(sel, return_inverse=True)
        # (reference: local indices come from the unique endpoints of the selected edges; a labelled vertex without a
        #  same-label edge is therefore not addressable -- the synthetic test graphs have none)
        L = laplacian(verts[inc], inv.reshape(sel.shape)) if sel.numel() else -torch.eye(int(inc.sum()))
        parts.append((int(label), inc, L))
    return parts


def region_laplacian_loss(x, verts, edges, labels, weights):
    loss = 0.
    for label, inc, L in _region_parts(verts, edges, labels):
        loss = loss + weights[label] * torch.matmul(L, x[inc]).pow(2).mean()
    return loss


def region_laplacian_hands(x, verts, edges, labels, hand_strength=1000):
    loss = 0.
    for label, inc, L in _region_parts(verts, edges, labels):
        if label in (6, 7):
            loss = loss + hand_strength * torch.matmul(L, x[inc]).pow(2).mean()
    return loss


def mesh_edge_loss(verts, edges):
    v0, v1 = verts[edges[:, 0]], verts[edges[:, 1]]
    return ((v0 - v1).norm(dim=1, p=2) ** 2).sum() / edges.shape[0]


def cot_laplacian(verts, faces, eps=1e-12):
    """pytorch3d.ops.cot_laplacian as a DENSE [V,V] matrix (the reference multiplies its sparse result with x: the same numbers)."""
    V = verts.shape[0]
    fv = verts[faces]
    v0, v1, v2 = fv[:, 0], fv[:, 1], fv[:, 2]
    A = (v1 - v2).norm(dim=1); B = (v0 - v2).norm(dim=1); C = (v0 - v1).norm(dim=1)
    s = 0.5 * (A + B + C)
    area = (s * (s - A) * (s - B) * (s - C)).clamp(min=eps).sqrt()
    A2, B2, C2 = A * A, B * B, C * C
    cot = torch.stack([(B2 + C2 - A2) / area, (A2 + C2 - B2) / area, (A2 + B2 - C2) / area], dim=1) / 4.0
    ii = faces[:, [1, 2, 0]].reshape(-1)
    jj = faces[:, [2, 0, 1]].reshape(-1)
    L = torch.zeros((V, V), dtype=verts.dtype)
    L.index_put_((ii, jj), cot.reshape(-1), accumulate=True)
    L = L + L.t()
    inv_areas = torch.zeros(V, dtype=verts.dtype)
    inv_areas.scatter_add_(0, faces.reshape(-1), torch.stack([area] * 3, dim=1).reshape(-1))
    nz = inv_areas > 0
    inv_areas[nz] = 1.0 / inv_areas[nz]
    return L, inv_areas.view(-1, 1)


def _region_parts_cot(verts, faces, labels):
    parts = []
    face_label = labels[faces]
    for label in torch.unique(labels):
        sel = faces[torch.any(face_label == label, dim=1)]
        inc = torch.unique(sel)                                  # global ids of the region's vertices (sorted)
        _, inv = torch.unique(sel, return_inverse=True)
        L, _ = cot_laplacian(verts[inc], inv.reshape(sel.shape))
        parts.append((int(label), inc, L))
    return parts


def region_laplacian_cot_loss(x, verts, faces, labels, weights, only=None, strength=None):
    """forward (``only`` None: every label, weights[label]) / forward_hands (``only`` = (6, 7), ``strength``)."""
    loss = 0.
    for label, inc, L in _region_parts_cot(verts, faces, labels):
        if only is not None and label not in only:
            continue
        w = strength if only is not None else weights[label]
        loss = loss + w * torch.matmul(L, x[inc]).pow(2).mean()
    return loss
