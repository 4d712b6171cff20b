#!/bin/bash
# usage: tools/e5abl.sh  -- C4 on the streaming kernel: ablation builds (staging only / staging + stores / full) on one box
one() { python bench.py --workload c4 --eval-kernel 5 --no-cpu --no-extra --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print('%-10s kernel_us=%7.2f frac=%.3f seg_rows=%d env=%s' % ('$TAG', r['kernel_us'], r['frac'], c['tile_rows'], {k: v for k, v in c['env'].items() if k != 'VARANNEAL_AMD_LIB'}))
"; }
for ns in 3 4; do
  for t in abl1 abl2; do TAG=$t VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_$t.so VA_E5_NSLOT=$ns one; done
  TAG=full VA_E5_NSLOT=$ns one
done
TAG=full VA_E5_NSLOT=4 VA_E5_XDPP=0 one
TAG=full VA_E5_NSLOT=3 one --tile-rows 250
TAG=full VA_E5_NSLOT=3 one --tile-rows 210
TAG=full VA_E5_NSLOT=3 one --tile-rows 158
TAG=full VA_E5_NSLOT=4 one --tile-rows 626
