"""GPU parity of the stand-alone lbs_extra: sings_amd.lbs.lbs_extra -- the reference's signature and return
values (sings/rec/utils/body_model/lbs.py:16-74) -- against golden G3 (the reference's own lbs_extra, tests/golden/
gen_lbs_golden.py) and, for gradients / batches / the pose-corrective path, the torch-CPU oracle in fp64."""
import os

import numpy as np
import pytest
import torch

from oracle import lbs_oracle as lo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "lbs_golden.npz"))


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("J", [24, 52])
def test_lbs_extra_golden(J):
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    A, w, v = t(G[f"g3_A_{J}"]), t(G[f"g3_w_{J}"]), t(G[f"g3_v_{J}"])
    verts, A_out, T, v_posed, v_shaped = lbs_extra(A[None], v[None], None, w, torch.zeros(1, J * 3, device=dev), disable_posedirs=True,
                                                   pose2rot=True)
    assert verts.shape == (1, 1000, 3) and T.shape == (1, 1000, 4, 4) and A_out is not None and v_posed.shape == (1, 1000, 3)
    # the reference sums the J (and 4) products in rocBLAS / MKL order, the kernel in joint order on the matrix cores
    np.testing.assert_allclose(T[0].cpu().numpy(), G[f"g3_T_{J}"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(verts[0].cpu().numpy(), G[f"g3_verts_{J}"], rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("N,J,B", [(3001, 52, 2), (64, 24, 1), (1, 3, 1), (777, 64, 1)])
def test_lbs_extra_gradients_and_batches(N, J, B):
    """verts, T and the gradients w.r.t. A and v (upstream gradients on BOTH outputs) against the fp64 oracle; N not a
    multiple of the 64-point waves, J not a multiple of 4 / 16, J = 64 (the maximum), a batch of frames."""
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    rs = np.random.RandomState(N + J)
    A_n = (np.eye(4)[None, None] + 0.3 * rs.randn(B, J, 4, 4)).astype(np.float32)
    w_n = rs.rand(N, J).astype(np.float32) ** 4; w_n /= w_n.sum(1, keepdims=True)
    v_n = rs.randn(B, N, 3).astype(np.float32)
    gv, gT = rs.randn(B, N, 3).astype(np.float32), rs.randn(B, N, 4, 4).astype(np.float32)
    A = torch.from_numpy(A_n).to(dev).requires_grad_(True); v = torch.from_numpy(v_n).to(dev).requires_grad_(True)
    verts, _, T, _, _ = lbs_extra(A, v, None, torch.from_numpy(w_n).to(dev), None, disable_posedirs=True)
    ((verts * torch.from_numpy(gv).to(dev)).sum() + (T * torch.from_numpy(gT).to(dev)).sum()).backward()
    A64 = torch.from_numpy(A_n).double().requires_grad_(True); v64 = torch.from_numpy(v_n).double().requires_grad_(True)
    vo, To = lo.lbs_extra(A64, v64, torch.from_numpy(w_n).double())
    ((vo * torch.from_numpy(gv).double()).sum() + (To * torch.from_numpy(gT).double()).sum()).backward()

    def close(a, b, what):
        a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().numpy()
        assert (np.abs(a - b) <= 2e-5 * np.abs(b) + 2e-6 * np.abs(b).max()).all(), (what, np.abs(a - b).max())
    close(verts, vo, "verts"); close(T, To, "T"); close(v.grad, v64.grad, "dv"); close(A.grad, A64.grad, "dA")
    # only one of the two outputs used downstream (the other gradient is None / zero)
    A.grad = None; v.grad = None
    verts2, _, T2, _, _ = lbs_extra(A, v, None, torch.from_numpy(w_n).to(dev), None, disable_posedirs=True)
    (T2 * torch.from_numpy(gT).to(dev)).sum().backward()
    A64.grad = None; v64.grad = None
    vo, To = lo.lbs_extra(A64, v64, torch.from_numpy(w_n).double())
    (To * torch.from_numpy(gT).double()).sum().backward()
    close(A.grad, A64.grad, "dA (T only)")
    assert float(v.grad.abs().max()) == 0.0


def test_lbs_extra_pose_correctives_and_errors():
    """disable_posedirs=False: v_posed = v_shaped + pose_feature @ posedirs (lbs.py:27-36) before the skinning."""
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    rs = np.random.RandomState(3)
    N, J = 500, 24
    A = torch.from_numpy((np.eye(4)[None, None] + 0.2 * rs.randn(1, J, 4, 4)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rs.rand(N, J).astype(np.float32)).to(dev); w = w / w.sum(1, keepdim=True)
    v = torch.from_numpy(rs.randn(1, N, 3).astype(np.float32)).to(dev)
    posedirs = torch.from_numpy((0.01 * rs.randn((J - 1) * 9, N * 3)).astype(np.float32)).to(dev)
    pose = torch.from_numpy((0.3 * rs.randn(1, J * 3)).astype(np.float32)).to(dev)
    verts, _, T, v_posed, v_shaped = lbs_extra(A, v, posedirs, w, pose, disable_posedirs=False, pose2rot=True)
    R = lo.batch_rodrigues(pose.cpu().double().view(-1, 3)).view(1, J, 3, 3)
    off = ((R[:, 1:] - torch.eye(3, dtype=torch.float64)).reshape(1, -1) @ posedirs.cpu().double()).view(1, N, 3)
    vo, To = lo.lbs_extra(A.cpu().double(), v.cpu().double() + off, w.cpu().double())
    np.testing.assert_allclose(v_posed.cpu().numpy(), (v.cpu().double() + off).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(verts.cpu().numpy(), vo.numpy(), rtol=2e-5, atol=2e-5)
    assert v_shaped is v
    with pytest.raises(RuntimeError):
        lbs_extra(A, v, None, w.clone().requires_grad_(True), None, disable_posedirs=True)
    with pytest.raises(RuntimeError):
        lbs_extra(A.cpu(), v.cpu(), None, w.cpu(), None, disable_posedirs=True)


def test_matrix_to_quaternion_golden_and_gradient():
    """sings_amd.rotations.matrix_to_quaternion on the GPU (sg_matrix_to_quaternion) against golden G1 -- the reference's own
    function, run on the CPU, on 256 seeded matrices incl. non-orthonormal ones: within one ulp (7 of 1024 words differ by the
    last bit, exactly where torch's own GPU kernels differ from its CPU kernels), BIT-identical to the reference expression
    evaluated by torch on the same GPU -- and its backward against torch autograd (all four candidate branches, the floor)."""
    from oracle import rotations_oracle as RO
    from sings_amd import rotations as R
    dev = _dev()
    m = torch.from_numpy(G["g1_mats"]).to(dev)
    q = R.matrix_to_quaternion(m)
    np.testing.assert_allclose(q.cpu().numpy(), G["g1_m2q"], rtol=1.3e-7, atol=0)
    assert torch.equal(q, RO.matrix_to_quaternion(m))
    rs = np.random.RandomState(4)
    big = torch.from_numpy(np.concatenate([G["g1_mats"], 0.05 * rs.randn(64, 3, 3).astype(np.float32),        # floor branch
                                           RO.axis_angle_to_matrix(torch.from_numpy(3.1 * rs.randn(2000, 3).astype(np.float32))).numpy()]))
    g = torch.from_numpy(rs.randn(big.shape[0], 4).astype(np.float32))
    a = big.to(dev).requires_grad_(True)
    (R.matrix_to_quaternion(a) * g.to(dev)).sum().backward()
    b = big.clone().requires_grad_(True)
    qb = RO.matrix_to_quaternion(b)
    (qb * g).sum().backward()
    assert len(set(np.abs(qb.detach().numpy()).argmax(1).tolist())) == 4               # every candidate branch is exercised
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=2e-6, atol=2e-6)
    # batch dimensions; host tensors are refused (no CPU path in the product)
    assert R.matrix_to_quaternion(m.view(16, 16, 3, 3)).shape == (16, 16, 4)
    with pytest.raises(RuntimeError):
        R.matrix_to_quaternion(m.cpu())


@pytest.mark.parametrize("J,B", [(24, 5), (52, 3), (64, 1), (1, 2)])
def test_joint_transforms_kernel_vs_torch_chain(J, B):
    """sg_joint_transforms (body.joint_transforms_hip): the kinematic chain of smpl.py:415-513 in one launch, against the torch
    restatement (pinned to golden G2 by tests/test_oracle_lbs.py) in fp64 -- values, gradients w.r.t. pose and rest joints,
    the optional per-joint right factor; random trees in topological order; zero poses (the 1e-8 of batch_rodrigues)."""
    from sings_amd import body
    dev = _dev()
    rs = np.random.RandomState(J * 7 + B)
    parents = (-1,) + tuple(int(rs.randint(0, i)) for i in range(1, J)) if J != 24 else body.SMPL_PARENTS
    pose_n = (0.6 * rs.randn(B, J * 3)).astype(np.float32); pose_n[0, :6] = 0.0
    jr_n = rs.randn(J, 3).astype(np.float32); post_n = (np.eye(4)[None] + 0.2 * rs.randn(J, 4, 4)).astype(np.float32)
    g_n = rs.randn(B, J, 4, 4).astype(np.float32)
    for use_post in (False, True):
        pose = torch.from_numpy(pose_n).to(dev).requires_grad_(True); jr = torch.from_numpy(jr_n).to(dev).requires_grad_(True)
        post = torch.from_numpy(post_n).to(dev) if use_post else None
        A = body.joint_transforms_hip(pose, jr, parents, post)
        (A * torch.from_numpy(g_n).to(dev)).sum().backward()
        p64 = torch.from_numpy(pose_n).double().requires_grad_(True); j64 = torch.from_numpy(jr_n).double().requires_grad_(True)
        ref = torch.stack([body._joint_transforms_torch(p64[b], j64, parents) for b in range(B)])
        if use_post:
            ref = ref @ torch.from_numpy(post_n).double()[None]
        (ref * torch.from_numpy(g_n).double()).sum().backward()

        def close(a, b, what):
            a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().numpy()
            assert (np.abs(a - b) <= 3e-5 * np.abs(b) + 3e-6 * np.abs(b).max()).all(), (what, use_post, np.abs(a - b).max())
        close(A, ref, "A"); close(pose.grad, p64.grad, "dpose"); close(jr.grad, j64.grad, "djoints")
    # the drop-in entry points route GPU fp32 tensors through the kernel and agree with the torch path
    one = body.joint_transforms(torch.from_numpy(pose_n[B - 1]).to(dev), torch.from_numpy(jr_n).to(dev), parents)
    np.testing.assert_allclose(one.cpu().numpy(), body._joint_transforms_torch(torch.from_numpy(pose_n[B - 1]), torch.from_numpy(jr_n), parents).numpy(),
                               rtol=2e-5, atol=2e-5)
    with pytest.raises(ValueError):
        body.joint_transforms_hip(torch.zeros(1, 9, device=dev), torch.zeros(3, 3, device=dev), (-1, 2, 0))


def test_joint_transforms_and_template_skinning_against_golden_g2():
    """SURVEY.md 8 row a10 on the HIP path, against the REFERENCE's numbers: golden G2 holds the outputs of the reference's own
    batch_rodrigues + batch_rigid_transform + lbs (sings/rec/utils/body_model/smpl.py:274-513) for the seeded SMPL-shaped model
    (V = 6 890, J = 24) at three poses (zero, the 'da' pose, one AMASS frame).
     * sg_joint_transforms (one launch for the three poses) vs G["g2_A_i"], and the posed joints A [j;1] + ... vs G["g2_J_i"];
     * sg_lbs_forward on the 6 890-vertex shaped template with those A (BASELINE configs[0]'s shape) vs G["g2_verts_i"];
     * sg_joint_transforms_backward (dpose, djoints) vs the fp64 autograd of oracle/lbs_oracle.py's restatement of the chain."""
    from sings_amd import body
    from sings_amd.lbs import lbs_extra
    dev = _dev()
    bm = lo.synthetic_body_model(seed=0)
    parents = tuple(int(p) for p in bm["parents"])
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    # shaped template and rest joints: the blend-shape / regressor einsums of smpl.py:371-412, once per shape, on the host
    v_shaped = T(bm["v_template"]) + torch.einsum("l,mkl->mk", T(G["g2_betas"][0]), T(bm["shapedirs"]))
    J_rest = torch.einsum("ik,ji->jk", v_shaped, T(bm["J_regressor"]))
    poses = T(G["g2_poses"])                                                   # [3, 72]
    pose_d = poses.to(dev).requires_grad_(True); jr_d = J_rest.to(dev).requires_grad_(True)
    A = body.joint_transforms_hip(pose_d, jr_d, parents)                        # [3, 24, 4, 4], ONE launch
    for i in range(3):
        np.testing.assert_allclose(A[i].detach().cpu().numpy(), G[f"g2_A_{i}"][0], rtol=0, atol=3e-6)
        # posed joints: G_j's translation = A_j [j_rest; 1]
        Ai = A[i].detach().cpu()
        posed = torch.einsum("jab,jb->ja", Ai[:, :3, :3], J_rest) + Ai[:, :3, 3]
        np.testing.assert_allclose(posed.numpy(), G[f"g2_J_{i}"][0], rtol=0, atol=3e-6)
        verts, _, Tm, _, _ = lbs_extra(A[i:i + 1].detach(), v_shaped[None].to(dev), None, T(bm["lbs_weights"]).to(dev), None,
                                       disable_posedirs=True)
        assert verts.shape == (1, 6890, 3)
        np.testing.assert_allclose(verts[0].cpu().numpy(), G[f"g2_verts_{i}"][0], rtol=0, atol=5e-6)
    assert np.abs(A[0].detach().cpu().numpy() - np.eye(4)).max() < 1e-5         # zero pose: identities (the 1e-8 of batch_rodrigues)
    # backward against the oracle's chain in fp64
    g_n = np.random.RandomState(2).randn(3, 24, 4, 4).astype(np.float32)
    (A * torch.from_numpy(g_n).to(dev)).sum().backward()
    p64 = poses.double().requires_grad_(True); j64 = J_rest.double().requires_grad_(True)
    R64 = lo.batch_rodrigues(p64.view(-1, 3)).view(3, 24, 3, 3)
    _, A64 = lo.batch_rigid_transform(R64, j64[None].expand(3, -1, -1), list(parents))
    np.testing.assert_allclose(A.detach().cpu().numpy(), A64.detach().numpy(), rtol=0, atol=3e-6)
    (A64 * torch.from_numpy(g_n).double()).sum().backward()

    def close(a, b, what):
        a = a.detach().cpu().numpy().astype(np.float64); b = b.detach().numpy()
        assert (np.abs(a - b) <= 3e-5 * np.abs(b) + 3e-6 * np.abs(b).max()).all(), (what, np.abs(a - b).max())
    close(pose_d.grad, p64.grad, "dpose"); close(jr_d.grad, j64.grad, "djoints_rest")
