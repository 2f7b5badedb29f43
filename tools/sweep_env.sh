#!/bin/bash
# Step time of the benchmark frame for several values of one environment knob:
#   tools/sweep_env.sh NAME "v1 v2 ..." [bench.py options]
name=$1; vals=$2; shift 2
for v in $vals; do
env $name=$v timeout -k 10 300 python3 -u bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=b.get('strong_scaling_proxy') or {}; print('$name=$v', b['ms_per_step'], p.get('part_0_of_8_ms'), b['kernel_wall_ms_per_frame_rank0'], flush=True)" || exit 1
done
