#!/bin/bash
# Step time of the benchmark frame with several builds of the library (tools/variant.sh), in one process each,
# on the same machine: tools/sweep_lib.sh "<name> <name> ..." [bench.py options]   ("-" = the product build)
names=$1; shift
for rep in 1 2; do for n in $names; do
lib=$PWD/raytracing_amd/librt_amd_$n.so; [ "$n" = "-" ] && lib=$PWD/raytracing_amd/librt_amd.so
RT_AMD_LIB=$lib timeout -k 10 300 python3 -u bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=b.get('strong_scaling_proxy') or {}; print('$n', b['ms_per_step'], p.get('part_0_of_8_ms'), b['kernel_wall_ms_per_frame_rank0'], flush=True)" || exit 1
done; done
