#!/bin/bash
# Development helper: frame time of the benchmark scene for several gear-change thresholds.
export PYTHONPATH=.
for live in ${LIVES:-0 8 16 32}; do
  for parts in 1 8; do
    echo "== RT_COOP_LIVE=$live parts=$parts overlap=${RT_OVERLAP:-0}"
    RT_COOP_LIVE=$live RT_PARTS=$parts RT_OVERLAP=${RT_OVERLAP:-0} timeout -k 10 100 python3 -u tools/sweep.py config3 26,12 || exit 1
  done
done
