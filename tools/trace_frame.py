"""Timeline of the last frame in a rocprofv3 kernel_trace.csv: start / end of every launch in microseconds.
Usage: python tools/trace_frame.py <output directory of rocprofv3 --kernel-trace>"""
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(p)) if "k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fold = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_fold")]
lo = fold[-2] + 1 if len(fold) > 1 else 0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:fold[-1] + 1]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{r['Kernel_Name'].split('(')[0].replace('void ', ''):18s} start {a/1e3:8.0f}  end {b/1e3:8.0f}  dur {(b-a)/1e3:7.0f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
