#!/bin/bash
# usage: tools/kbench.sh "<bench args>" [env assignments...]  -- one line: kernel, kernel_us, frac
args=$1; shift
env "$@" python bench.py --no-cpu --no-extra --steps 500 --warmup 50 $args 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('%-44s kernel_us=%7.2f frac=%.3f ms_per_step=%.4f  [$args $*]' % (r['kernel'], r['kernel_us'], r['frac'], j['ms_per_step']))
"
