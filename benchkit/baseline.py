"""bench.py: the CPU baseline leg (PyTorch-CPU LBS + project in a child process, the scalar C raster oracle) and the parity log that
compares the oracle's full-size views with the engine's.  The ONLY benchkit module that touches oracle/ (test infrastructure: never timed
as the product, never on the product path)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

from .distrib import (COMM_KEYS, FORCE_DIST, LIGHT_TIMED_S, MAX_REPEATS, MIN_TIMED_S, _grad_sha256, _log, _median, _ranks_agree,   # noqa: F401
                      allreduce_probe, densification_stats_check, dp_self_check, exposed_by_algorithm, make_frame_parallel,
                      one_view_step_by_algorithm, timed_region, timed_repeats, usable_cores)
from .roofline import (HBM_COPY_GBS, HBM_PEAK_GBS, ROOT, algorithmic_bytes, algorithmic_bytes_skinned, build_roofline, measure_copy_peak,   # noqa: F401
                       pmc_view_traffic, scaling_model, train_step_roofline)


def cpu_lbs_project_worker(argv):
    """Child process of the cpu_baseline leg (never touches the GPU): PyTorch-CPU "LBS + project" with `threads` threads,
    median of 10 runs at N = 6 890 / 50 k / 200 k (+ the workload's own N); prints one JSON line.  Runs in a child so that the
    parent can bound it with a timeout (an over-subscribed OpenMP team can take minutes per call).  `kind`: "raster" -- the first
    N Gaussians of the benchmark scene with seeded sparse J = 52 skinning weights and near-identity joint transforms (the
    arithmetic does not depend on their values); "avatar" -- the avatar scene's own canonical points, weights and an AMASS pose."""
    import math
    import numpy as np
    import torch
    from oracle import lbs_project_torch as lp
    threads, kind, Ntot, W, H, deg = int(argv[0]), argv[1], *(int(v) for v in argv[2:6])
    torch.set_num_threads(threads)
    T = torch.from_numpy
    J = 52
    if kind == "avatar":
        from oracle import lbs_oracle as lo
        from sings_amd.scene import avatar_scene
        s = avatar_scene(N=Ntot, J=J)
        cam = s["cam"]
        poses72 = np.load(os.path.join(ROOT, "tests", "golden", "lbs_golden.npz"))["amass_poses_72"]
        pose = np.zeros(J * 3, np.float32); pose[:72] = poses72[0]; pose[:3] = 0
        R = lo.batch_rodrigues(T(pose).view(-1, 3)).view(1, J, 3, 3)
        A = lo.batch_rigid_transform(R, T(s["joints_rest"])[None], list(s["parents"]))[1][0]
        base = (s["xyz_canon"], s["scales"], s["opacities"], s["shs"], s["lbs_weights"])
        tail = (A, T(s["smpl_scale"]), T(s["transl"]), T(cam["world_view_transform"]), T(cam["full_proj_transform"]),
                T(cam["camera_center"]), s["W"], s["H"], math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5))
    else:
        from sings_amd.scene import synthetic_scene
        s = synthetic_scene(Ntot, W, H, deg, 3)
        rsd = np.random.RandomState(11)
        w = np.zeros((Ntot, J), np.float32)
        ja, jb = rsd.randint(0, J, Ntot), rsd.randint(0, J, Ntot)
        u = rsd.rand(Ntot).astype(np.float32)
        w[np.arange(Ntot), ja] = u; w[np.arange(Ntot), jb] += 1 - u
        A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1)); A[:, :3, 3] = rsd.normal(0, 1e-3, (J, 3))
        base = (s["means3D"], s["scales"], s["opacities"], s["shs"], w)
        tail = (T(A), torch.ones(1), torch.zeros(3), T(s["viewmatrix"]), T(s["projmatrix"]), T(s["campos"]), W, H, s["tanfovx"],
                s["tanfovy"])
    sweep = {}
    for n in dict.fromkeys((6890, 50000, 200000, Ntot)):
        idx = np.arange(n) % Ntot
        xyz, sc, op, sh, w_ = (T(np.ascontiguousarray(x[idx])) for x in base)
        args = (xyz, torch.eye(3)[None].repeat(n, 1, 1), sc, op, sh, deg, w_) + tail
        lp.lbs_project(*args)                                   # (first call: thread pool start-up)
        ts = []
        for _ in range(10):
            t1 = time.perf_counter(); lp.lbs_project(*args); ts.append(time.perf_counter() - t1)
        sweep[str(n)] = round(sorted(ts)[len(ts) // 2] * 1e3, 3)
    print(json.dumps({"threads": threads, "median_ms_by_points": sweep, "torch": torch.__version__}), flush=True)


def cpu_lbs_project(kind, Ntot, W, H, deg):
    """SURVEY.md 8(d) / BASELINE.md section 3: PyTorch-CPU "LBS + project" -- skinning (W.A, T[v;1]), rotation compose,
    matrix_to_quaternion, then cull / project / cov3D / cov2D / radius / SH -- on the host cores this process may use
    (`usable_cores`, stated), in a child process under a timeout; if the full team does not finish (over-subscription) the
    16-thread figure is reported and the line says so.  ONE implementation for every workload.  -> the cpu_baseline dict."""
    cores = usable_cores()
    tried, res = [], None
    for threads in dict.fromkeys((cores, min(cores, 16))):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--_cpu-worker", str(threads), kind, str(Ntot), str(W),
                                str(H), str(deg)], capture_output=True, text=True, timeout=150)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and line:
                res = json.loads(line[-1])
                tried.append({"threads": threads, "ok": True})
                break
            tried.append({"threads": threads, "ok": False, "rc": p.returncode, "stderr": p.stderr[-300:]})
        except subprocess.TimeoutExpired:
            tried.append({"threads": threads, "ok": False, "timeout_s": 150})
    cpu_model = ""
    try:
        cpu_model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:
        pass
    if res is None:
        return {"value": None, "unit": "frames/s", "cores": cores, "kind": "port", "sample": "PyTorch-CPU LBS + project did not finish",
                "attempts": tried, "host_cpus": os.cpu_count(), "usable_cores": cores}
    ms = res["median_ms_by_points"][str(Ntot)] if str(Ntot) in res["median_ms_by_points"] else res["median_ms_by_points"]["200000"]
    n_at = Ntot if str(Ntot) in res["median_ms_by_points"] else 200000
    return {"value": 1e3 / ms, "unit": "frames/s (PyTorch-CPU LBS + project only: no binning, no composite, no backward)",
            "cores": res["threads"], "kind": "port",
            "sample": f"oracle/lbs_project_torch.py on the {kind} scene, J=52, SH deg {deg}, median of 10 runs per size, N={n_at}: {ms} ms",
            "median_ms_by_points": res["median_ms_by_points"], "cpu_model": cpu_model, "torch": res["torch"],
            "host_cpus": os.cpu_count(), "usable_cores": cores, "attempts": tried}


def cpu_baseline(s, camera, deg, W, H, gpu_view=None, n_views=3, lbs_project=True, backward=True):
    """The raster workloads' cpu_baseline + parity: `cpu_lbs_project` (the reported baseline), and the scalar C restatement of the
    whole rasterizer (1 core, `n_views` full views fwd+bwd of this run's cameras: a bounded sample) riding along as an extra key --
    its images and gradients are COMPARED with the engine's for the same cameras (`gpu_view(v, dL) -> dict`).
    -> (cpu_baseline, parity)."""
    from oracle import raster_oracle as ro
    Ntot = s["means3D"].shape[0]
    tc = 0.0
    par = _ParityLog()
    for v in range(n_views):
        v_, p_, c_, _ = camera(v)
        t0 = time.perf_counter()
        o = ro.forward(s["means3D"], s["opacities"], v_, p_, c_, W, H, s["tanfovx"], s["tanfovy"], s["bg"],
                       scales=s["scales"], rotations=s["rotations"], shs=s["shs"], sh_degree=deg, want_margin=True)
        # pixels whose hard-threshold decisions are borderline in the oracle carry no loss, on both sides (tests/test_gpu_raster.py)
        dLn = s["dL_dimage"].copy(); dLn[:, o["margin"] < PARITY_BORDER] = 0
        g = ro.backward(o, dLn) if backward else None
        tc += time.perf_counter() - t0
        if gpu_view is not None:                                 # the checker's result is USED: the engine's view v against it
            par.add(o, g, gpu_view(v, dLn))
    raster = {"value": n_views / tc, "unit": "views/s", "cores": 1, "kind": "port",
              "sample": f"{n_views} full view(s) {'fwd+bwd' if backward else 'forward'} of the same scene with the scalar C oracle "
                        f"({tc:.1f} s; the forward also computes the per-pixel threshold margins the parity block needs)"}
    parity = par.result() if gpu_view is not None else None
    if not lbs_project:
        return raster, parity
    cb = cpu_lbs_project("raster", Ntot, W, H, deg)
    cb["raster_oracle_1core"] = raster
    return cb, parity


PARITY_BORDER = 2e-5          # tests/test_gpu_raster.py::BORDER
PARITY_RGB_TOL = 1e-5         # BASELINE.json north_star: per-pixel RGB within 1e-5 of the reference


class _ParityLog:
    """HIP engine vs CPU oracle over the full-size views of the cpu_baseline leg -- the bars of tests/test_gpu_raster.py
    (bit-exact binning; RGB <= 1e-5 on pixels whose threshold decisions have a margin, borderline ones within the oracle's own
    flip bound; every gradient within rtol 2e-4 + 2e-6 of the array's scale), applied to the benchmark's own configuration on
    every run.  Outside the timed region; the oracle is the checker, never the thing measured as `value`."""

    def __init__(self):
        self.views = 0
        self.binning_exact = True
        self.rgb_linf = 0.0
        self.border_px = 0
        self.border_beyond_tol = 0
        self.border_beyond_flip = 0
        self.grad_max_rel = 0.0
        self.grad_violations = 0
        self.grads_compared = 0
        self.failed = []

    def add(self, o, g, d):
        import numpy as np
        self.views += 1
        if d.get("error"):
            self.failed.append(d["error"]); self.binning_exact = False
            return
        exact = (d["R"] == o["R"] and np.array_equal(d["radii"], o["radii"]) and
                 np.array_equal(d["ranges"].astype(np.uint32), o["ranges"]) and
                 np.array_equal(d["point_list"].astype(np.uint32), o["point_list"]))
        self.binning_exact = self.binning_exact and bool(exact)
        diff = np.abs(d["color"] - o["color"]).max(0)
        border = o["margin"] < PARITY_BORDER
        self.rgb_linf = max(self.rgb_linf, float(diff[~border].max()))
        self.border_px += int(border.sum())
        if border.any():
            self.border_beyond_tol += int((diff[border] > PARITY_RGB_TOL).sum())
            self.border_beyond_flip += int((diff[border] > PARITY_RGB_TOL + 1.001 * o["flip"][border]).sum())
        if g is None:
            return
        for name, key in (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmean2D"), ("opacity", "dL_dopacity"), ("scales", "dL_dscales"),
                          ("rotations", "dL_drots"), ("sh", "dL_dsh")):
            self.add_grad(d["grads"][name], g[key])

    def add_grad(self, a, b, rtol=2e-4, atol=2e-6):
        import numpy as np
        b = np.asarray(b, np.float64); a = np.asarray(a, np.float64).reshape(b.shape)
        scale = np.abs(b).max() + 1e-30
        err = np.abs(a - b)
        self.grads_compared += 1
        self.grad_max_rel = max(self.grad_max_rel, float(err.max() / scale))
        self.grad_violations += int((err > rtol * np.abs(b) + atol * scale).sum())

    def result(self):
        ok = (self.binning_exact and self.rgb_linf <= PARITY_RGB_TOL and self.border_beyond_flip == 0 and self.grad_violations == 0
              and not self.failed)
        return {"views": self.views, "ok": bool(ok), "binning_exact": bool(self.binning_exact), "rgb_linf": self.rgb_linf,
                "rgb_tol": PARITY_RGB_TOL, "borderline_px": self.border_px, "borderline_px_beyond_1e-5": self.border_beyond_tol,
                "borderline_px_beyond_flip_bound": self.border_beyond_flip, "grad_max_rel": self.grad_max_rel,
                "grad_violations": self.grad_violations, "gradient_arrays_compared": self.grads_compared, "grad_tol": "rtol 2e-4 + 2e-6 x max|g| per array (tests/test_gpu_raster.py)",
                "errors": self.failed,
                "against": "oracle/raster_oracle (scalar C restatement, fp32; PARITY UNPINNED: DESIGN.md section 2), full-size views "
                           "of this run's cameras 0..views-1, R / radii / ranges / point_list compared bit for bit"}
