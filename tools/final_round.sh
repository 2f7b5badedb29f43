#!/bin/bash
# Everything a round's profiles/ are refreshed from, in one GPU call: the GPU tests, tools/profile_round.sh <tag>,
# and the bench lines of the other two workloads.   tools/final_round.sh <tag>
TAG=${1:-r03}
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/$TAG/gputests.log 2>&1 || { tail -5 gpurun_out/$TAG/gputests.log; exit 1; }
tail -1 gpurun_out/$TAG/gputests.log
tools/profile_round.sh $TAG || exit 1
timeout -k 10 300 python3 -u bench.py --workload config2 --depth 22 --leaf 12 > gpurun_out/$TAG/bench_config2.json 2> gpurun_out/$TAG/bench_config2.err || exit 1
timeout -k 10 400 python3 -u bench.py --workload config5 > gpurun_out/$TAG/bench_config5.json 2> gpurun_out/$TAG/bench_config5.err || exit 1
# counter passes of the fly-through frame (incoherent secondary rays): lane utilisation, wait ratio, L2 hit rate
RT_LAUNCH_WAVES=8 tools/pmc_passes.sh gpurun_out/$TAG/pmc_config5 config5 26 12 2 || exit 1
echo final_round done
