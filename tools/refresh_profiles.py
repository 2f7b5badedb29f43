"""Copies the results of tools/profile_round.sh <tag> from gpurun_out/<tag>/ into profiles/:
kernel stats, counter summaries, the per-kernel fabric traffic table and the two bench lines.
Usage: python tools/refresh_profiles.py <tag> [workload/depthD/leafL/gpusN]"""
import csv, glob, json, os, shutil, subprocess, sys

tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "config3/depth26/leaf12/gpus1"
src = os.path.join("gpurun_out", tag)
shutil.copy(glob.glob(src + "/stats/*/*_kernel_stats.csv")[0], "profiles/r01_kernel_stats_final.csv")
shutil.copy(src + "/bench.json", "profiles/r01_bench_config3.json")
shutil.copy(src + "/bench_under_rocprof.json", "profiles/r01_bench_config3_under_rocprof.json")


def summary(d, out):
    subprocess.run([sys.executable, "tools/pmc_summary.py", d, out], check=True, stdout=subprocess.DEVNULL)
    per = {}
    for r in csv.DictReader(open(out)):
        k = r["kernel"].replace("void ", "").split("<")[0]
        b = (2 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024     # counters are in KiB
        per.setdefault(k, [0.0, 0])
        per[k][0] += b
        per[k][1] += int(r["dispatches"])
    t = {k: float("%.7g" % (v[0] / v[1])) for k, v in per.items()}
    t["k_shade"] = t.get("k_light")
    return t


tr = json.load(open("profiles/r01_traffic.json"))
tr[key + "/overlap"] = summary(src + "/pmc", "profiles/r01_pmc_summary_final.csv")
tr[key + "/serial"] = summary(src + "/pmc_serial", "/tmp/pmc_serial.csv")
json.dump(tr, open("profiles/r01_traffic.json", "w"), indent=1)
b = json.load(open("profiles/r01_bench_config3.json"))
print({k: b[k] for k in ("value", "ms_per_step", "kernel_ms_per_frame_rank0", "kernel_ms_per_frame_isolated_rank0")})
print(b["roofline"])
print(tr[key + "/overlap"])
