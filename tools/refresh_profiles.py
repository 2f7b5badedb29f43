"""Copies the results of tools/profile_round.sh <tag> from gpurun_out/<tag>/ into profiles/ under the
round's prefix: kernel stats, counter summaries, the two bench lines, and <round>_counters.json, the per-kernel
table bench.py quotes (fabric traffic per launch and the ratios that show what binds a kernel).
Usage: python tools/refresh_profiles.py <tag> <round> [workload/depthD/leafL[/pruned]/gpusN]"""
import csv, glob, json, os, shutil, subprocess, sys

tag, rnd = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "config3/depth26/leaf12/pruned/gpus1"
src = os.path.join("gpurun_out", tag)
def newest(pattern):
    """(results of several runs under one tag pile up in gpurun_out: the latest one counts)"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


shutil.copy(newest(src + "/stats/*/*_kernel_stats.csv")[0], f"profiles/{rnd}_kernel_stats.csv")
ser = newest(src + "/stats_serial/*/*_kernel_stats.csv")
if ser:
    shutil.copy(ser[0], f"profiles/{rnd}_kernel_stats_serial.csv")
    shutil.copy(src + "/bench_serial_under_rocprof.json", f"profiles/{rnd}_bench_config3_serial_under_rocprof.json")
shutil.copy(src + "/bench.json", f"profiles/{rnd}_bench_config3.json")
shutil.copy(src + "/bench_under_rocprof.json", f"profiles/{rnd}_bench_config3_under_rocprof.json")


def summary(d, out):
    subprocess.run([sys.executable, "tools/pmc_summary.py", d, out], check=True, stdout=subprocess.DEVNULL)
    per = {}
    for r in csv.DictReader(open(out)):
        k = r["kernel"].replace("void ", "").split("<")[0]
        acc = per.setdefault(k, {})
        for name, val in r.items():
            if name != "kernel" and val != "":
                acc[name] = acc.get(name, 0.0) + float(val)
    table = {}
    for k, c in per.items():
        g = lambda n: c.get(n, 0.0)
        n = max(g("dispatches"), 1.0)
        sig = lambda x: float("%.5g" % x)
        table[k] = {
            "dispatches": int(n),
            # FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half the bytes fetched
            "fabric_bytes_per_launch": sig((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / n),
            "fabric_read_bytes_per_launch": sig(2 * g("FETCH_SIZE") * 1024 / n),
            "fabric_write_bytes_per_launch": sig(g("WRITE_SIZE") * 1024 / n),
            "wait_any_over_wave_cycles": sig(g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1.0)),
            "lane_utilisation": sig(g("SQ_THREAD_CYCLES_VALU") / max(64.0 * g("SQ_ACTIVE_INST_VALU"), 1.0)),
            "l2_hit_rate": sig(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1.0)),
            "valu_insts_per_launch": sig(g("SQ_INSTS_VALU") / n),
            "vmem_read_insts_per_launch": sig(g("SQ_INSTS_VMEM_RD") / n),
            "lds_insts_per_launch": sig(g("SQ_INSTS_LDS") / n),
            "lds_bank_conflict_over_wave_cycles": sig(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_WAVE_CYCLES"), 1.0)),
            "l1_read_latency_cycles": sig(g("TCP_TCC_READ_REQ_LATENCY_sum") / max(g("TCP_TCC_READ_REQ_sum"), 1.0)),
        }
    if "k_light" in table:
        table["k_shade"] = table["k_light"]
    return table


path = f"profiles/{rnd}_counters.json"
tr = json.load(open(path)) if os.path.exists(path) else {}
tr[key + "/overlap"] = summary(src + "/pmc", f"profiles/{rnd}_pmc_summary.csv")
tr[key + "/serial"] = summary(src + "/pmc_serial", f"profiles/{rnd}_pmc_summary_serial.csv")
if os.path.isdir(src + "/pmc_config5"):
    tr[key.replace("config3", "config5") + "/overlap"] = summary(src + "/pmc_config5", f"profiles/{rnd}_pmc_summary_config5.csv")
for w in ("config2", "config5"):
    if os.path.exists(f"{src}/bench_{w}.json"):
        shutil.copy(f"{src}/bench_{w}.json", f"profiles/{rnd}_bench_{w}.json")
json.dump(tr, open(path, "w"), indent=1)
b = json.load(open(f"profiles/{rnd}_bench_config3.json"))
print({k: b[k] for k in ("value", "ms_per_step", "kernel_ms_per_frame_rank0", "kernel_wall_ms_per_frame_rank0")})
print(b["roofline"])
print(json.dumps(tr[key + "/overlap"], indent=1))
