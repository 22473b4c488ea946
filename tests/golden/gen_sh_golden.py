"""Generates tests/golden/sh_golden.npz (G9) by EXECUTING the reference's own spherical-harmonics file in the build container:
sings/rec/utils/visualize/spherical_harmonics.py (eval_sh :54-113, constants :28-47) -- the one in-tree statement of the SH basis
the un-vendored rasterizer evaluates in its preprocess.  The file moves its constants to the
GPU at import time (`.cuda()` x5); there is no GPU here, so those five calls are stripped from the text before it is executed --
nothing else is touched (the way gen_render_glue_golden.py handles `device="cuda"`).  The reference source never ships; only the
vectors do.

    python tests/golden/gen_sh_golden.py

Stored: 512 seeded unit directions (every octant; grouped by dominant axis so that a test can put each group in front of a camera),
SH coefficients [512,16,3] in the rasterizer's layout, eval_sh for degrees 0..3 ([512,3] each, WITHOUT the +0.5 / clamp the
rasterizer adds), and the basis values d eval_sh / d sh [512,16] at degree 3 (autograd through the reference function).
"""
import linecache
import os
import types

import numpy as np
import torch

SRC = "/root/reference/sings/rec/utils/visualize/spherical_harmonics.py"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

src = open(SRC).read()
assert src.count(".cuda()") == 5
ref = types.ModuleType("ref_spherical_harmonics")
text = src.replace(".cuda()", "")
FN = "<reference spherical_harmonics.py, .cuda() stripped>"
linecache.cache[FN] = (len(text), None, text.splitlines(True), FN)      # @torch.jit.script reads the function's source through inspect
exec(compile(text, FN, "exec"), ref.__dict__)

rs = np.random.RandomState(99)
n = 512
d = rs.normal(size=(n, 3))
d /= np.linalg.norm(d, axis=1, keepdims=True)
d[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)      # the axes themselves
d = d.astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)                          # unit in fp32 as far as fp32 goes
sh = np.concatenate([rs.normal(0, 1, (n, 1, 3)), rs.normal(0, 0.5, (n, 15, 3))], 1).astype(np.float32)

out = {"dirs": d, "sh": sh}
consts = (ref.C0, ref.C1, ref.C2, ref.C3, ref.C4)
sh_t = torch.from_numpy(sh).transpose(1, 2).contiguous()              # reference layout [..., C, coefficients]
for deg in range(4):
    out[f"eval_deg{deg}"] = ref.eval_sh(deg, sh_t, torch.from_numpy(d), *consts).numpy()
# basis values: d result / d sh_k (the same for the three channels) by autograd through the reference function
one = torch.zeros(n, 1, 16, requires_grad=True)
ref.eval_sh(3, one, torch.from_numpy(d), *consts).sum().backward()
out["basis_deg3"] = one.grad[:, 0, :].numpy()
out["C0"] = np.float32(ref.C0.item())
assert abs(float(ref.SH2RGB(torch.tensor(1.0))) - (0.28209479177387814 + 0.5)) < 1e-7

dst = os.path.join(ROOT, "tests", "golden", "sh_golden.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst) // 1024, "KiB", {k: getattr(v, "shape", None) for k, v in out.items()})
