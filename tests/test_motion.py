"""f2: motion-sequence preparation (sings_amd/motion.py) against golden G7 -- the REFERENCE's rebase_smpl / manual_alignment
run by tests/golden/gen_motion_golden.py -- on the CPU; forward_chunk (golden G8) on the GPU."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "motion_golden.npz"))


def test_rebase_smpl_matches_the_reference_including_its_quirks():
    from sings_amd import motion
    poses, transl = torch.from_numpy(G["g7_poses"]), torch.from_numpy(G["g7_transl"])
    p, t = motion.rebase_smpl(poses, transl)
    assert p is poses                                            # the reference returns `poses` untouched
    np.testing.assert_array_equal(p.numpy(), G["g7_poses_out"])
    assert t.shape == (15, 3, 1) and t.dtype == torch.float32    # [N,3,1], as the reference's matmul leaves it
    np.testing.assert_allclose(t.numpy(), G["g7_transl_out"], rtol=0, atol=2e-6)
    assert abs(float(t[0, 2, 0]) - 20.0) < 1e-6 and float(t[0, :2].abs().max()) < 1e-6
    p2, t2 = motion.rebase_smpl(poses.numpy(), transl.numpy())  # numpy in -> tensor out
    np.testing.assert_array_equal(t2.numpy(), t.numpy())


def test_manual_alignment_table():
    from sings_amd import motion
    for kind in ("AMASS", "custom", "other"):
        tr, ro, sc = motion.manual_alignment(kind)
        np.testing.assert_allclose(np.concatenate([tr, ro, [sc]]), G[f"g7_align_{kind}"], rtol=0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("iso", [False, True])
@pytest.mark.parametrize("ext", [False, True])
def test_forward_chunk_against_the_reference_pieces(iso, ext):
    """posed.forward_chunk == sings_hybrid.py:512-553 evaluated with the reference's own lbs_extra / rotations (G8)."""
    from sings_amd.posed import forward_chunk
    dev = torch.device("cuda:0")
    T = lambda k: torch.from_numpy(G["g8_" + k]).to(dev)
    attrs = dict(xyz_canon=T("xyz_canon"), xyz_offsets=None, rot6d_canon=T("rot6d"), scales=T("scales"),
                 opacity=torch.rand(500, 1, device=dev), shs=torch.rand(500, 16, 3, device=dev))
    out = forward_chunk(attrs, T("w"), T("A"), None, transl=T("transl"), smpl_scale=T("smpl_scale"),
                        ext_tfs=(T("ext_trans"), T("ext_rot"), T("ext_scale")) if ext else None, isotropic=iso, active_sh_degree=2)
    tag = f"g8_{'iso' if iso else 'aniso'}_{'ext' if ext else 'plain'}"
    np.testing.assert_allclose(out["xyz"].cpu().numpy(), G[tag + "_xyz"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["scales"].cpu().numpy(), G[tag + "_scales"], rtol=1e-6, atol=1e-9)
    q, qr = out["rotq"].cpu().numpy(), G[tag + "_rotq"]
    # matrix_to_quaternion picks the best-conditioned of four candidates: a different pick is the same rotation
    same = np.abs(q - qr).max(-1) < 2e-5
    assert same.mean() > 0.995, same.mean()
    assert out["xyz"].shape == (3, 500, 3) and out["rotq"].shape == (3, 500, 4) and out["shs"].shape == (3, 500, 16, 3)
    assert out["scales_canon"] is out["scales"] and out["active_sh_degree"] == 2
    if not iso:
        np.testing.assert_allclose(out["rotq_canon"].cpu().numpy(), G["g8_rotq_canon"], rtol=2e-5, atol=2e-6)
    else:
        assert float(out["rotq_canon"].abs().max()) == 0.0     # the reference's isotropic branch returns zeros here
