#!/bin/bash
# register / LDS / spill table of every kernel in a build of the library: tools/kernel_resources.sh [lib.so] [name filter]
LIB=${1:-brats2019_amd/lib/libresunet_hip.so}; PAT=${2:-.}
T=$(mktemp -d); cp "$LIB" $T/l.so
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading l.so >/dev/null 2>&1)
for f in $T/l.so.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f; done | python3 -c "
import sys,re
name=None; rec={}
for line in sys.stdin:
    m=re.search(r'^\s+(?:- )?\.(\w+):\s+(.*)$', line)
    if not m: continue
    k,v=m.group(1),m.group(2).strip()
    if k=='name' and v.startswith('_Z') or (k=='name' and 'kernel' in v): name=v
    if k in ('vgpr_count','sgpr_count','vgpr_spill_count','sgpr_spill_count','group_segment_fixed_size','private_segment_fixed_size','agpr_count'): rec[k]=v
    if k=='wavefront_size' and name:
        print('v%s s%s vsp%s ssp%s lds%s scr%s  %s' % (rec.get('vgpr_count'), rec.get('sgpr_count'), rec.get('vgpr_spill_count'), rec.get('sgpr_spill_count'), rec.get('group_segment_fixed_size'), rec.get('private_segment_fixed_size'), name)); name=None; rec={}
" | (command -v c++filt >/dev/null && c++filt || cat) | grep -E "$PAT"
rm -rf $T
