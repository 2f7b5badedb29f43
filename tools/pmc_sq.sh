#!/bin/bash
# Instruction-mix counters per dispatch. Usage: tools/pmc_sq.sh <outdir> [workload depth leaf frames]
set -u
OUT=${1:-gpurun_out/pmc_sq}
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
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU || exit 1
echo all passes done
