"""Scan the gfx950 ISA of every kernel for loops that serialise their memory latency: an innermost loop whose body waits with
s_waitcnt vmcnt(0) although it issues only one or two loads (load -> wait -> use per iteration: one request in flight per wave).
hipcc falls into this form for plain copy loops and whenever a load sits behind a branch.  python tools/isa_scan.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-fast-math", f"-I{ROOT}/include", "-S", "--cuda-device-only"]


def scan(path):
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", f.name, path], check=True, stderr=subprocess.DEVNULL)
        lines = open(f.name).read().splitlines()
    kern, out = None, []
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+|sg_\w+):", ln)
        if m:
            kern = m.group(1)
        if "Inner Loop Header" in ln:
            label = lines[i - 1].split(":")[0] if lines[i - 1].startswith(".LBB") else (ln.split(":")[0] if ln.startswith(".LBB") else None)
            # body = until the branch back to the label
            j, loads, waits0, atom = i + 1, 0, 0, 0
            while j < len(lines) and j < i + 400:
                t = lines[j].strip()
                if re.match(r"(global|buffer|flat|scratch)_load", t):
                    loads += 1
                if re.match(r"(global|buffer)_atomic", t):
                    atom += 1
                if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                    waits0 += 1
                if label and re.match(r"s_cbranch\w+ " + re.escape(label) + r"\b", t):
                    break
                if t.startswith(".Lfunc_end"):
                    break
                j += 1
            if loads and waits0 and loads <= 2 * waits0 + 1:
                out.append((kern, label, i + 1, loads, atom, waits0, j - i))
        i += 1
    return out


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "sings_amd", "csrc", "*.hip")))
    for p in files:
        for kern, label, line, loads, atom, waits0, length in scan(p):
            m = re.match(r"_Z(\d+)", kern)                      # (no c++filt in the image: the length-prefixed name is enough)
            name = kern[2 + len(m.group(1)):2 + len(m.group(1)) + int(m.group(1))] + kern[2 + len(m.group(1)) + int(m.group(1)):][:24] if m else kern
            print(f"{os.path.basename(p):18s} {name[:60]:60s} loop {label} (asm line {line}, {length} lines): {loads} loads, {atom} atomics, "
                  f"{waits0} x vmcnt(0)")
