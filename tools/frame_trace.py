"""Development helper: per-launch durations of the last plain frame in a rocprofv3 kernel trace
(tools/prof_kernels.sh output directory)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"].split("(")[0][:30], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
stats = [i for i, n in enumerate(names) if "<true" in n[0]]
end = stats[0] if stats else len(names)
folds = [i for i, n in enumerate(names[:end]) if n[0].startswith("k_fold")]
a, b = (folds[-2] + 1 if len(folds) > 1 else 0), folds[-1] + 1
t0 = names[a][1]
for n, s, e in names[a:b]:
    print(f"   {n:32s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us")
print(f"   frame {(names[b - 1][2] - t0) / 1e3:.1f} us")
