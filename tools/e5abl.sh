#!/bin/bash
# usage: tools/e5abl.sh  -- C4 on the streaming kernel, measurement builds on one box (python -m varanneal_amd._build --variant e5ablN -DVA_E5_ABLATE=N):
#   1 the staging ring alone | 2 ring + the gradient stores of a copy | 3 the ring alone, every wave reading ONE contiguous run | full
one() { python bench.py --workload c4 --no-cpu --no-extra --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print('%-10s kernel_us=%7.2f seg_rows=%d' % ('$TAG', r['kernel_us'], c['tile_rows']))
"; }
for rep in 1 2; do
  for v in 1 2 3; do
    so=$PWD/varanneal_amd/libvaranneal_amd_e5abl$v.so
    [ -f $so ] && TAG=abl$v VARANNEAL_AMD_LIB=$so one "$@"
  done
  TAG=full one "$@"
done
