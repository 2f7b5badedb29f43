#!/bin/bash
# Address-translation (UTCL1) counters of the L1 for the benchmark frame.
# Usage: tools/pmc_tlb.sh <outdir> [workload depth leaf frames]
set -u
OUT=${1:-gpurun_out/pmc_tlb}
shift || true
ARGS=${*:-config3 26 12 2}
export PYTHONPATH=.
export TMPDIR=/tmp
mkdir -p "$OUT"
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/bench_small.py $ARGS > "$OUT/$name.log" 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run tlb1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum GRBM_GUI_ACTIVE || exit 1
run tlb2 TCP_TCP_LATENCY_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum SQ_INSTS_VMEM_RD SQ_WAVES || exit 1
echo all passes done
