#!/bin/bash
# Step time of the benchmark frame over kd-tree limits: tools/sweep_tree.sh "<depths>" "<leaf sizes>"
for d in ${1:-26}; do for l in ${2:-12}; do
timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --no-proxy --depth $d --leaf $l 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($d, $l, b['ms_per_step'], b['roofline']['per_ray'], b['config']['kd_tree']['triangle_refs'], b['config']['kd_tree']['nodes'], flush=True)" || exit 1
done; done
