#!/bin/bash
# Registers and scratch of the frame kernels at 4 and 5 waves per SIMD, compiled from the working tree (no library is built):
#   tools/regs_try.sh [extra hipcc flags]
D=$(dirname "$0")/..
for w in 4 5; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -I$D/include -I$D/raytracing_amd/csrc --offload-arch=gfx950 -DRT_WAVES_CLOSEST=$w -DRT_WAVES_SHADOW=$w "$@" \
    -x hip --cuda-device-only -S $D/raytracing_amd/csrc/rt_hip.hip -o /tmp/regs_try_$w.s 2>/dev/null || { echo "compile failed"; exit 1; }
  echo "== $w waves per SIMD"
  awk '/[.]amdhsa_kernel /{k=$2} /\.amdhsa_next_free_vgpr/{v=$2} /\.amdhsa_private_segment_fixed_size/{p=$2} /\.end_amdhsa_kernel/{ if (k ~ /k_closestILb0|k_shadowILb0|k_light/) print "  ", substr(k,1,44), "vgpr", v, "scratch", p }' /tmp/regs_try_$w.s
done
