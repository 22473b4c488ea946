"""CPU: on-disk formats: text PLY layout of vis.py:22-61, .splat conversion of convert.py:11-50."""
import os

import numpy as np

from sings_amd import export

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "export_golden.npz"))


def _gs():
    return {k: G[k] for k in ("xyz_canon", "shs", "opacity", "scales_canon", "rotq_canon")}


def test_ply_layout_and_round_trip(tmp_path):
    p = str(tmp_path / "sub" / "human.ply")
    export.save_ply(_gs(), p)
    head = open(p).read().split("end_header\n")[0].splitlines()
    assert head[:3] == ["ply", "format ascii 1.0", "element vertex 257"]
    props = [l.split()[-1] for l in head[3:]]
    assert props == export.construct_list_of_attributes() and len(props) == 62
    assert props[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9] == "f_rest_0" and props[53] == "f_rest_44" and props[54] == "opacity"
    assert props[55:] == ["scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    v = export.load_ply(p)
    gs = _gs()
    np.testing.assert_array_equal(np.stack([v["x"], v["y"], v["z"]], 1), gs["xyz_canon"])          # %.9g round-trips fp32
    assert not np.any(v["nx"]) and not np.any(v["ny"]) and not np.any(v["nz"])
    np.testing.assert_array_equal(np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], 1), gs["shs"][:, 0, :])
    # channel-major SH rest: f_rest_0..14 = red coefficients 1..15, then green, then blue (vis.py:48)
    np.testing.assert_array_equal(v["f_rest_0"], gs["shs"][:, 1, 0])
    np.testing.assert_array_equal(v["f_rest_14"], gs["shs"][:, 15, 0])
    np.testing.assert_array_equal(v["f_rest_15"], gs["shs"][:, 1, 1])
    np.testing.assert_array_equal(v["f_rest_44"], gs["shs"][:, 15, 2])
    op = gs["opacity"][:, 0]
    np.testing.assert_array_equal(v["opacity"], np.log(op / (1 - op)))
    np.testing.assert_array_equal(np.stack([v["scale_0"], v["scale_1"], v["scale_2"]], 1), np.log(gs["scales_canon"]))
    np.testing.assert_array_equal(np.stack([v[f"rot_{i}"] for i in range(4)], 1), gs["rotq_canon"])
    assert bytes(G["ply_text_head"]) == open(p, "rb").read(4096)


def test_splat_bytes_match_reference_converter(tmp_path):
    p = str(tmp_path / "human.ply")
    export.save_ply(_gs(), p)
    v = export.load_ply(p)
    ref = bytes(G["splat_bytes"])                         # reference function, run under numpy 2 (NEP 50)
    ours = export.ply_to_splat(v, legacy_promotion=False)
    assert len(ours) == 32 * 257 and ours == ref
    # the reference's pinned numpy 1.23.5 promotes the colour arithmetic to float64: same order, same geometry bytes,
    # colour bytes within one step
    leg = np.frombuffer(export.ply_to_splat(v), dtype=np.uint8).reshape(-1, 32)
    r = np.frombuffer(ref, dtype=np.uint8).reshape(-1, 32)
    assert np.array_equal(leg[:, :24], r[:, :24]) and np.array_equal(leg[:, 28:], r[:, 28:])
    assert np.abs(leg[:, 24:28].astype(int) - r[:, 24:28].astype(int)).max() <= 1


def test_binary_ply_reader_and_checkpoint_keys(tmp_path):
    v = np.zeros(5, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v["x"] = np.arange(5); v["red"] = 7
    p = tmp_path / "lvl.ply"
    with open(p, "wb") as f:                               # layout of save_ply_by_level (vis.py:74-88), binary
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty float x\nproperty float y\n"
                b"property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        f.write(v.tobytes())
    w = export.load_ply(str(p))
    assert np.array_equal(w["x"], v["x"]) and np.array_equal(w["red"], v["red"])
    k = export.checkpoint_keys(num_gs_level=2)
    assert k[:3] == ["active_sh_degree", "xyz", "triplane"] and k[-4:] == ["appearance_dec_0", "geometry_dec_0", "appearance_dec_1", "geometry_dec_1"]


def test_morton_order_is_a_spatially_coherent_permutation():
    from sings_amd.scene import morton_order
    rs = np.random.RandomState(0)
    x = rs.uniform(-1, 1, (20000, 3)).astype(np.float32)
    p = morton_order(x)
    assert sorted(p.tolist()) == list(range(20000))
    step_sorted = np.linalg.norm(np.diff(x[p], axis=0), axis=1).mean()
    step_random = np.linalg.norm(np.diff(x, axis=0), axis=1).mean()
    assert step_sorted < 0.15 * step_random                  # neighbours in the order are neighbours in space
