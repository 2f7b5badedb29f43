"""Per-kernel times of the last device build in a rocprofv3 kernel trace of tools/build_prof.py
(rocprofv3 --kernel-trace --stats -d DIR -o bp -- python tools/build_prof.py; python tools/build_trace.py DIR/bp_results.db)."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x from kernels order by start"))
# the last build starts at the last k_extents
first = max(i for i, r in enumerate(rows) if "k_extents" in r[0])
last = rows[first:]
def short(nm):
    m = re.search(r"(k_\w+|init_lookback\w*|trampoline\w*|copyBuffer|fillBuffer\w*)", nm)
    return m.group(1) if m else nm[:60]
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for nm, s, e, g in last:
    agg[short(nm)][0] += 1; agg[short(nm)][1] += e - s; busy += e - s
print(f"last build: {len(last)} kernels, first start to last end {(last[-1][2] - last[0][1]) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{t / 1e3:9.1f} us {c:4d} {k}")
for name in sys.argv[2:]:
    print(name, [round((e - s) / 1e3, 1) for nm, s, e, g in last if name in nm])
if "--timeline" in sys.argv:
    prev = last[0][1]
    for nm, s, e, g in last:
        print(f"{(s - last[0][1]) / 1e3:8.1f} +{(s - prev) / 1e3:6.1f} gap {(e - s) / 1e3:7.1f} us  {short(nm)} grid {g}")
        prev = e
