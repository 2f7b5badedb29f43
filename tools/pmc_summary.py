"""Summarise rocprofv3 --pmc CSV output per kernel: sums each counter over the
dispatches of every kernel.  Usage: python tools/pmc_summary.py <dir> [out.csv]"""
import csv, glob, os, sys, collections
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace(", ", ";")      # no commas inside a CSV field
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
names = sorted({c for k in tot for c in tot[k]})
rows = [["kernel", "dispatches"] + names]
for k in sorted(tot):
    if not any(x in k for x in ("k_closest", "k_shadow", "k_spawn", "k_light", "k_fold")):
        continue
    rows.append([k, max(calls[k].values())] + [("%.6g" % tot[k][c]) if c in tot[k] else "" for c in names])
out = "\n".join(",".join(str(x) for x in r) for r in rows)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
