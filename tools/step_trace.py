"""One graph-replayed training step as a timeline: start, duration, hardware queue and name of every kernel, and the idle gaps.
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o tt -- python3 bench.py --workload train --steps 10 --warmup 3
    python3 tools/step_trace.py <dir>/.../tt_kernel_trace.csv > profiles/<tag>_train_step_trace.log
(kernels of different streams overlap: the durations do not add up to the step)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "sg_triplane_fwd_kernel" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]                        # a step from the middle of the timed region
seg = rows[a - 3:b - 3]                              # (three launches precede the tri-plane forward: concat, copy, plane transpose)
t0 = int(seg[0]["Start_Timestamp"])
print(f"# one step of {len(starts)}: {len(seg)} kernels, {(max(int(r['End_Timestamp']) for r in seg) - t0) / 1e3:.1f} us from first start to last end")
last_end, gaps = t0, []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > last_end:
        gaps.append(((s - last_end) / 1e3, r["Kernel_Name"][:50]))
    last_end = max(last_end, e)
print(f"# nothing running: {sum(g for g, _ in gaps):.1f} us in {len(gaps)} gaps; largest: {sorted(gaps, reverse=True)[:5]}")
print("# start_us duration_us queue kernel")
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r['Kernel_Name'] if len(sys.argv) > 2 and sys.argv[2] == "--full" else r['Kernel_Name'].split('(')[0][:70]
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '?')} {name[:400]}")
