#!/bin/bash
# Profiles of one round, written under gpurun_out/<tag>/ on the GPU box:
#   bench.json / bench.err     the default bench.py line
#   stats/                     rocprofv3 --kernel-trace --stats of the same command
#   stats_serial/              the same of one frame at a time with the passes in order (isolated launch durations)
#   pmc/                       counter passes (one rocprofv3 run each) over a short run of the same frame
#   pmc_serial/                the same with every pass alone on the chip (RT_OVERLAP=0)
# Usage: tools/profile_round.sh <tag> ; then, here: python tools/refresh_profiles.py <tag> <round, e.g. r02>
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p $OUT
timeout -k 10 500 python3 -u bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo bench done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --timed-only > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
echo stats done
# the same frame, one at a time, every pass alone on the chip with all 16 waves per CU: the profiler's AverageNs of a
# kernel is then the duration of a launch, and algorithmic bytes per launch / AverageNs is the bench line's achieved_isolated
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 bench.py --no-cpu-baseline --timed-only --no-overlap --in-flight 1 --launch-waves 16 > $OUT/bench_serial_under_rocprof.json 2> $OUT/stats_serial.err || exit 1
echo serial stats done
export RT_LAUNCH_WAVES=8      # the counter passes render one frame at a time; give the passes the grids of the timed region
tools/pmc_passes.sh $OUT/pmc || exit 1
RT_OVERLAP=0 tools/pmc_passes.sh $OUT/pmc_serial || exit 1
echo profile_round done
