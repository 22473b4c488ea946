"""python tests/tools/seam_ulps.py -- prints the posed-value errors of the fused LBS kernel against oracle/lbs_oracle.py in ulps
(the numbers behind the bounds of tests/test_gpu_skinned.py::test_posed_values_seam_in_ulps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_skinned as T
dev = torch.device("cuda:0")
for name, s in (("generic J=52", T._scene(6000, 52, 2)), ("generic J=24", T._scene(6000, 24, 1)), ("generic J=30", T._scene(6000, 30, 4)),
                ("avatar-shaped J=52", T._avatar_shaped())):
    print(name, "ulps (xyz, quaternion, scales):", T.seam_ulps(s, dev), flush=True)
