#!/bin/bash
# usage: tools/kstat.sh <object built by _build.py> [name filter]  -- registers / spills / LDS of the gfx950 kernels in it
obj=$1; pat=${2:-.}
tmp=$(mktemp -d); B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
$B/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/k.co
$B/llvm-readelf --notes $tmp/k.co | python3 -c "
import sys,re
cur={}
for ln in sys.stdin:
    m=re.match(r'\s*(- )?\.(\w+):\s*(.*)',ln)
    if not m: continue
    k,v=m.group(2),m.group(3).strip()
    if k=='name' and v.startswith('_Z'):
        cur={'name':v}
    elif k in ('vgpr_count','sgpr_count','vgpr_spill_count','sgpr_spill_count','private_segment_fixed_size','group_segment_fixed_size'): cur[k]=v
    if k=='vgpr_spill_count' and 'name' in cur and re.search(r'$pat',cur['name']):
        print('%-90s vgpr=%s sgpr=%s vspill=%s sspill=%s scratch=%s' % (cur['name'][:90],cur.get('vgpr_count'),cur.get('sgpr_count'),cur.get('vgpr_spill_count'),cur.get('sgpr_spill_count'),cur.get('private_segment_fixed_size')))
"
rm -rf $tmp
