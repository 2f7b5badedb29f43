#!/bin/bash
# Texture-addresser / data-return / L1 busy counters for the benchmark frame.
# Usage: tools/pmc_ta.sh <outdir> [workload depth leaf frames]
set -u
OUT=${1:-gpurun_out/pmc_ta}
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
run ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
run ta2 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum || exit 1
echo all passes done
