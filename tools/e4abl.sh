#!/bin/bash
# usage: tools/e4abl.sh [bench args]  -- C3 on k_eval4: ablation builds on one box (profiles/r03_ablation_c3.txt)
#   1 the launch alone | 2 + staging (LDS-DMA, wait) | 3 + gradient stores of a copy | 4 rows phase + stores, no gather, no tail
#   5 + gather phase, no tail | full
one() { python bench.py --no-cpu --no-extra --steps 3000 --warmup 300 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('%-8s kernel_us=%6.3f ms_per_step=%.5f' % ('$TAG', r['kernel_us'], j['ms_per_step']))
"; }
for rep in 1 2; do
for v in 1 2 3 4 5; do TAG=abl$v VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_e4abl$v.so one "$@"; done
TAG=full one "$@"
done
