"""Diagnostic (GPU box): clock and workgroup lifetimes of sg_render_bwd_kernel with ONE view per launch against 8 cameras per launch.
Needs a library built with -DSG_RENDER_STAMP:  SINGS_HIP_LIB=build/lib_rstamp.so python tools/render_stamps.py"""
import os, sys, ctypes as C, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1:                                                   # child: run the bench form, then dump the stamps of its LAST launch
    import bench
    sys.argv = ["bench.py"] + sys.argv[1:]
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    from sings_amd import _lib
    lib = _lib.load()
    n = 65536 * 4
    arr = (C.c_ulonglong * n)()
    lib.sg_debug_render_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.sg_debug_render_stamps(arr, n) == 0
    s = np.frombuffer(arr, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    s = s[s[:, 3] > 0]
    c0, r0, c1, r1 = s.T
    t0 = r0.min()
    life = (r1 - r0) / 100.0
    clk = (c1 - c0) / np.maximum(r1 - r0, 1) * 100.0
    long_ = life > np.percentile(life, 50)
    j = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(json.dumps({"workgroups": int(len(s)), "span_us": float((r1.max() - t0) / 100.0), "start_p50_us": float(np.median((r0 - t0) / 100.0)),
                      "start_max_us": float(((r0 - t0) / 100.0).max()), "life_median_us": float(np.median(life)), "life_max_us": float(life.max()),
                      "clock_MHz_median_of_long_workgroups": float(np.median(clk[long_])), "kernel_ms_bwd": j["kernel_ms"]["sg_render_bwd_kernel"]}))
    sys.exit(0)
for name, args in (("one view per launch", ["--views-per-step", "1", "--streams", "1"]),
                   ("8 cameras per launch", ["--views-per-step", "8", "--frames-per-launch", "8", "--streams", "1"])):
    p = subprocess.run([sys.executable, __file__, "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"] + args,
                       capture_output=True, text=True)
    print(name, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:])
