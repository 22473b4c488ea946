"""Per-view kernel timeline from a rocprofv3 --kernel-trace CSV of `bench.py --views-per-step 1 --streams 1`:
for every kernel of a view (in launch order) its mean duration and the mean GAP between the end of the kernel before it
and its own start -- kernel boundaries are a tenth of a 350-us view, invisible in `--stats`.
    python tools/timeline.py <..._kernel_trace.csv> [out.csv] [must=<kernel name prefix>]
must=...: only steps that contain such a kernel (a K-frame run -- `--frames-per-launch 8 --streams 1` -- also holds the one-view
steps of the bench's second leg: must=sg_record_sums_kernel picks the K-frame steps; durations are then per K frames)."""
import csv, re, sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
views, cur = [], []
for st, en, name in rows:
    if not name.startswith("sg_"):
        continue
    if name.startswith(("sg_zero_kernel", "sg_preprocess_fwd_kernel", "sg_skin_fwd_kernel")) and cur and \
            not (name.startswith(("sg_preprocess_fwd_kernel", "sg_skin_fwd_kernel")) and cur[-1][2].startswith("sg_zero_kernel")):
        views.append(cur); cur = []
    cur.append((st, en, name))
if cur:
    views.append(cur)
# keep the views of the dominant shape (forward + backward of the timed loops), drop warm-up / sizing / profiling oddities
shape = defaultdict(int)
for v in views:
    shape[tuple(k[2] for k in v)] += 1
must = [a[5:] for a in sys.argv[3:] if a.startswith("must=")]
if must:
    shape = {k: n for k, n in shape.items() if any(x.startswith(must[0]) for x in k)}
best = max(shape, key=shape.get)
sel = [v for v in views if tuple(k[2] for k in v) == best][5:]
out = []
tot_d = tot_g = 0.0
for i, name in enumerate(best):
    d = sum(v[i][1] - v[i][0] for v in sel) / len(sel) / 1e3
    g = sum(v[i][0] - v[i - 1][1] for v in sel) / len(sel) / 1e3 if i else 0.0
    out.append((i, name, d, g)); tot_d += d; tot_g += g
span = sum(v[-1][1] - v[0][0] for v in sel) / len(sel) / 1e3
period = sum(sel[j + 1][0][0] - sel[j][0][0] for j in range(len(sel) - 1)) / max(len(sel) - 1, 1) / 1e3
lines = ["position,kernel,mean_duration_us,mean_gap_before_us"] + [f"{i},{n},{d:.2f},{g:.2f}" for i, n, d, g in out]
lines.append(f"#views,{len(sel)},sum_durations_us,{tot_d:.2f},sum_gaps_us,{tot_g:.2f},first_start_to_last_end_us,{span:.2f},view_period_us,{period:.2f}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
