"""Generates tests/golden/export_golden.npz: the bytes the reference's own ``process_ply_to_splat``
(playground/display/convert.py:11-50) produces for a PLY written by sings_amd.export.save_ply.

    python tests/golden/gen_export_golden.py

convert.py reads the file through the third-party ``plyfile`` package (not installed here); a placeholder module
whose ``PlyData.read`` returns our own parser's structured array stands in for it -- the conversion code under test
is the reference's, unchanged.  NOTE: this container has numpy 2.2 (NEP 50 promotion), the reference pins 1.23.5.
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sings_amd import export                                      # noqa: E402

pf = types.ModuleType("plyfile")


class PlyData(dict):
    @staticmethod
    def read(path):
        return PlyData(vertex=export.load_ply(path))


pf.PlyData = PlyData
sys.modules["plyfile"] = pf
sys.path.insert(0, "/root/reference/playground/display")
import convert as ref_convert                                     # noqa: E402

rs = np.random.RandomState(11)
N = 257
gs = dict(xyz_canon=rs.normal(0, 0.5, (N, 3)).astype(np.float32), shs=rs.normal(0, 0.8, (N, 16, 3)).astype(np.float32),
          opacity=rs.uniform(0.01, 0.99, (N, 1)).astype(np.float32),
          scales_canon=np.exp(rs.normal(-4, 1, (N, 3))).astype(np.float32), rotq_canon=rs.normal(0, 1, (N, 4)).astype(np.float32))
path = "/tmp/export_golden.ply"
export.save_ply(gs, path)
splat = ref_convert.process_ply_to_splat(path)
out = {k: v for k, v in gs.items()}
out["splat_bytes"] = np.frombuffer(splat, dtype=np.uint8)
out["ply_text_head"] = np.frombuffer(open(path, "rb").read(4096), dtype=np.uint8)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "export_golden.npz"), **out)
print("wrote export_golden.npz:", len(splat), "splat bytes")
