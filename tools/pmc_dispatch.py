"""Per-dispatch counter table of the last frame in rocprofv3 counter_collection.csv files.
Usage: python tools/pmc_dispatch.py <dir with pass sub-directories>"""
import csv, glob, sys, collections

def main():
    root = sys.argv[1]
    table = collections.defaultdict(dict)      # dispatch order -> counter -> value
    names = {}
    for p in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(p)) if "k_" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        order = {d: i for i, d in enumerate(ids)}
        for r in rows:
            i = order[int(r["Dispatch_Id"])]
            table[i][r["Counter_Name"]] = table[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[i] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    fold = [i for i in sorted(names) if names[i].startswith("k_fold")]
    lo = fold[-2] + 1 if len(fold) > 1 else 0
    cols = sorted({c for i in table for c in table[i]})
    print("kernel".ljust(18) + " ".join(c[-22:].rjust(23) for c in cols))
    for i in range(lo, fold[-1] + 1):
        print(names[i][:18].ljust(18) + " ".join(f"{table[i].get(c, 0):23.4g}" for c in cols))

main()
