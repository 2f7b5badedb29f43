"""Development helper: how full the GPU is during a traced run.  From a rocprofv3 kernel trace (directory of
tools/prof_kernels.sh or any --kernel-trace run): over the last `frac` of the run, the fraction of wall time with
at least one frame kernel running, the mean number of kernels running at once, and the per-kernel share of
kernel-seconds.  Usage: python tools/trace_concurrency.py <dir> [frac=0.5]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [(r["Kernel_Name"].split("(")[0][:28], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows = [r for r in rows if any(k in r[0] for k in ("k_closest", "k_shadow", "k_light", "k_fold"))]
rows.sort(key=lambda r: r[1])
t_end = max(r[2] for r in rows)
t0 = rows[0][1] + (t_end - rows[0][1]) * (1.0 - frac)
rows = [r for r in rows if r[1] >= t0]
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
busy = area = 0
depth, last = 0, ev[0][0]
hist = collections.Counter()
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    hist[depth] += t - last
    depth += d
    last = t
wall = ev[-1][0] - ev[0][0]
per = collections.Counter()
for n, s, e in rows:
    per[n] += e - s
folds = sum(1 for r in rows if r[0].startswith("k_fold"))
print(f"window {wall / 1e6:.2f} ms, {folds} frames -> {wall / 1e6 / max(folds, 1):.3f} ms per frame; some kernel running {busy / wall:.3f} of the time; "
      f"mean kernels running {area / wall:.2f}")
print("time share by number of kernels running:", {k: round(v / wall, 3) for k, v in sorted(hist.items())})
print("kernel-ms per frame:", {k: round(v / 1e6 / max(folds, 1), 3) for k, v in per.items()})
