#!/bin/bash
# Frames in flight against streams per context, for rank 0's share of an E-way split (tools/inflight.py):
#   tools/inflight_sweep.sh <E> <out> ["K overlap waves" ...]
E=${1:-8}; OUT=${2:-gpurun_out/inflight_sweep.log}; shift 2
export PYTHONPATH=. GPU_MAX_HW_QUEUES=${QUEUES:-16} SHARE=1 DOUT=1
mkdir -p $(dirname $OUT); : > $OUT
for cfg in "$@"; do
  set -- $cfg
  RT_LAUNCH_WAVES=$3 timeout -k 10 120 python3 -u tools/inflight.py $1 $E $2 2>/dev/null | grep "in flight" | sed "s/^/waves $3: /" >> $OUT || exit 1
done
sed -E 's/, timing.*: ([0-9.]+ ms per frame)/ -> \1/' $OUT
