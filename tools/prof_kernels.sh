#!/bin/bash
# Development helper: per-kernel times (rocprofv3 --kernel-trace --stats) of a few benchmark frames.
# Usage: tools/prof_kernels.sh <out dir> [sweep.py arguments]; environment (RT_PARTS, RT_OVERLAP, RT_DEFER_*) is passed on.
OUT=$1; shift
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/sweep.py "$@" > $OUT/run.log 2>&1 || exit 1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
