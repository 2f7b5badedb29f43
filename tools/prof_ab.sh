#!/bin/bash
# Chunk-profiling builds (tools/variant.sh <name> <source> -DRT_PROFILE_CHUNKS) side by side: per-slot round / expand
# figures of the benchmark frame and of a 1/8 share, passes in order.   tools/prof_ab.sh "<name> <name> ..."
export PYTHONPATH=.
for n in $1; do for parts in 1 8; do
echo "== $n parts=$parts"
RT_AMD_LIB=$PWD/raytracing_amd/librt_amd_$n.so RT_PARTS=$parts RT_OVERLAP=0 timeout -k 10 120 python3 -u tools/sweep.py config3 26,12 2>&1 | grep -E "prof|frame" || exit 1
done; done
