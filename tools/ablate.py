"""Timing-only ablation builds of ONE kernel source (results are WRONG by construction: never loaded by the product).
    python tools/ablate.py <spec.py>        spec: SOURCE = "sg_skin.hip"; VARIANTS = {"name": [(old, new) | ("header.h", old, new), ...], ...}
-> build/exp/lib_<name>.so (the patched object linked with the tree's other objects).  On the GPU box:
    SINGS_HIP_LIB=build/exp/lib_<name>.so python bench.py ...
The product sources carry no experiment switches: the patches live in the spec file, outside the library."""
import os, runpy, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sings_amd", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
spec = runpy.run_path(sys.argv[1])
src = spec["SOURCE"]
objs = [os.path.join(CSRC, f[:-4] + ".o") for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and f != src]
os.makedirs(os.path.join(ROOT, "build", "exp"), exist_ok=True)
text = open(os.path.join(CSRC, src)).read()
procs = []
extra = []
for name, reps in spec["VARIANTS"].items():
    t = text
    hdrs = {}                                                  # a 3-tuple (header, old, new) patches a private copy of that header
    for rep in reps:
        if len(rep) == 3:
            h, old, new = rep
            ht = hdrs.get(h) or open(os.path.join(CSRC, h)).read()
            assert old in ht, (name, h, old[:80])
            hdrs[h] = ht.replace(old, new, 1)
            continue
        old, new = rep
        assert old in t, (name, old[:80])
        t = t.replace(old, new, 1)
    for h, ht in hdrs.items():
        hp = os.path.join(CSRC, f"_abl_{name}_{h}")
        open(hp, "w").write(ht)
        extra.append(hp)
        assert f'#include "{h}"' in t
        t = t.replace(f'#include "{h}"', f'#include "_abl_{name}_{h}"', 1)
    tmp = os.path.join(CSRC, f"_abl_{name}.hip")              # (next to the headers it includes; removed below)
    open(tmp, "w").write(t)
    obj = os.path.join(ROOT, "build", "exp", f"{name}.o")
    procs.append((name, tmp, obj, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", tmp, "-o", obj])))
for name, tmp, obj, p in procs:
    rc = p.wait()
    os.remove(tmp)
    if rc:
        raise SystemExit(f"variant {name} failed to compile")
    so = os.path.join(ROOT, "build", "exp", f"lib_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs)
    os.remove(obj)
    print("built", os.path.relpath(so, ROOT))
for hp in extra:
    os.remove(hp)
