#!/bin/bash
# Runs the CPU tests of the host code (kd builder, OBJ / MTL loader, host math, render-pipe bookkeeping)
# and the OBJ fuzzers against the sanitizer build of the host sources (make -C raytracing_amd/csrc asan).
# The sanitizer runtime has to be the first library of the process, hence the preload.
set -e
cd "$(dirname "$0")/.."
make -s -C raytracing_amd/csrc asan
export RT_AMD_LIB=$PWD/raytracing_amd/librt_amd_asan.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)${LD_PRELOAD:+:$LD_PRELOAD}
export RT_NO_LIVE_REFERENCE=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 PYTHONPATH=$PWD
python3 -m pytest tests/test_kdtree.py tests/test_mesh_loader.py tests/test_host_math.py tests/asan/test_pipe_bookkeeping.py -q -x -p no:cacheprovider "$@"
python3 tools/fuzz_obj.py 0 120 malformed
echo "sanitizer run clean"
