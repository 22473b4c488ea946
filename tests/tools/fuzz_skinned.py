"""Randomised sweep of the fused LBS + raster path against the oracles (GPU box): python tests/tools/fuzz_skinned.py [n_cases]
Re-runs tests/test_gpu_skinned.py::test_fused_forward / test_fused_backward with other seeds, joint counts and both rotation modes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_skinned as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "4321")))
for c in range(n):
    J = int(rs.choice([24, 30, 52])); iso = bool(rs.rand() < 0.3); seed = int(rs.randint(1 << 20))
    T.test_fused_forward(J, iso, seed)
    T.test_fused_backward(J, iso, seed + 1)
    print(f"case {c}: J={J} isotropic={iso} seeds {seed}, {seed + 1}  ok", flush=True)
print("all cases passed")
