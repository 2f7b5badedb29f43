#!/bin/bash
# Counter passes over a short run of the benchmark frame (one rocprofv3 run
# per pass; SQ <= 8, TCC <= 4 counters per pass).  Usage: tools/pmc_passes.sh <outdir> [workload depth leaf]
set -u
OUT=${1:-gpurun_out/pmc}
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
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM || exit 1
run tcc1 FETCH_SIZE TCC_HIT_sum || exit 1
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1
run tcc3 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum || exit 1
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
echo all passes done
