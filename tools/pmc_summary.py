"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files -> profiles/<tag>_pmc_<set>.csv and profiles/hbm_traffic.json.
python tools/pmc_summary.py gpurun_out/prof_r01d r01d
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1 KiB (MI355X_MICROARCH.md "HBM": gfx950 tallies 128-B fabric reads as
64 B; both counters are in KiB)."""
import csv, json, os, re, sys
from collections import defaultdict
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
# What the passes profiled: the bench configuration (default: cfg3, the command of tools/collect_profiles.sh; otherwise
# `workload=avatar gaussians=150000 width=512 height=896 sh_degree=0` style arguments) and the git blob hashes of the kernel
# sources AS THEY ARE in the tree the passes ran from -- run this script in that tree.  bench.py refuses counts whose sidecar
# does not match the run (bench._committed_pmc).
import bench
config = {"workload": "raster", "gaussians": 200000, "width": 1920, "height": 1080, "sh_degree": 3}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    config[k] = v if k == "workload" else int(v)
# frames_per_launch=K: the passes ran `--views-per-step K --frames-per-launch K --streams 1` (K frames / cameras per dispatch; the
# GPU-side reducer of tools/collect_profiles.sh keeps the K-frame dispatches only): the means are divided by K -> per VIEW
frames = int(config.get("frames_per_launch", 1))
sched = f"--views-per-step {frames} --frames-per-launch {frames} --streams 1" if frames > 1 else "--views-per-step 1 --streams 1"
meta = {"config": config, "sources": bench.source_hashes(root),
        "command": f"rocprofv3 --pmc <set> -- python3 bench.py --steps 5 --warmup 2 {sched} --no-cpu-baseline"
                   + ("" if config["workload"] == "raster" else f" --workload {config['workload']}")}


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


means = {}
for cset in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2"):
    f = os.path.join(src, f"pmc_{cset}", "c3_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = defaultdict(lambda: [0.0, 0])
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("sg_", "void sg_"))]
    names = {short(r["Kernel_Name"]).split("<")[0] for r in rows}
    # not part of a view: the bandwidth probe of the bench line; and, in a K-frame run, the kernels only its one-view-per-step leg
    # launches (their K-frame forms have their own names)
    skip = {"sg_copy_probe_kernel"}
    if frames > 1:
        for k1, kk in (("sg_render_fwd_kernel", "sg_render_fwd_frames_kernel"), ("sg_render_fwd_deep_kernel", "sg_render_fwd_frames_kernel"),
                       ("sg_preprocess_bwd_kernel", "sg_preprocess_bwd_frames_kernel"), ("sg_skin_bwd_kernel", "sg_skin_bwd_frames_kernel")):
            if kk in names:
                skip.add(k1)
    for r in rows:
        if short(r["Kernel_Name"]).split("<")[0] in skip:
            continue
        a = acc[(short(r["Kernel_Name"]), r["Counter_Name"])]
        n = int(r.get("Count") or 1)                       # (tools/collect_profiles.sh leaves per-(kernel, counter) means + counts)
        a[0] += float(r["Counter_Value"]) * n / frames; a[1] += n
    out = os.path.join(root, "profiles", f"{tag}_pmc_{cset}.csv")
    with open(out, "w") as fo:
        fo.write("kernel,Counter_Name,mean,count\n")
        for (k, c), (s, n) in sorted(acc.items()):
            fo.write(f'"{k}",{c},{s / n:.1f},{n}\n')            # (template arguments contain commas)
            means[(k, c)] = s / n
    json.dump(meta, open(out[:-4] + ".meta.json", "w"), indent=1, sort_keys=True)
    print("wrote", out, "+ .meta.json")
traffic = {}
for (k, c) in list(means):
    if c == "FETCH_SIZE" and (k, "WRITE_SIZE") in means:
        traffic[k] = int((2 * means[(k, "FETCH_SIZE")] + means[(k, "WRITE_SIZE")]) * 1024)
if traffic:
    traffic["_meta"] = meta
    # the headline configuration keeps the historical name; every other configuration gets its own file (bench.py picks the
    # one whose "_meta" matches the run)
    default_cfg = {"workload": "raster", "gaussians": 200000, "width": 1920, "height": 1080, "sh_degree": 3}
    name = "hbm_traffic.json" if config == default_cfg else f"{tag}_hbm_traffic.json"      # (K-frame passes: tag them, e.g. r04_k8)
    json.dump(traffic, open(os.path.join(root, "profiles", name), "w"), indent=1, sort_keys=True)
    print(f"wrote profiles/{name}", {k: v for k, v in traffic.items() if k != "_meta"})
