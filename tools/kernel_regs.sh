#!/bin/bash
# Registers, scratch and code size of the frame kernels inside a built library:
#   tools/kernel_regs.sh [library]
LIB=${1:-raytracing_amd/librt_amd.so}
T=$(mktemp -d); cp "$LIB" $T/lib.so
( cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null )
for o in $T/lib.so.*gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $o | python3 -c "
import re,sys
t=sys.stdin.read()
for blk in t.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    n=g('name')
    if re.search('k_closest|k_shadow|k_light|k_fold',n): print(n[:60], 'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'))
"
  /opt/rocm/lib/llvm/bin/llvm-readelf -s $o | awk '/k_closest|k_shadow/ && /FUNC/ {print $8, $3}'
done
rm -rf $T
