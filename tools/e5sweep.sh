#!/bin/bash
# usage: tools/e5sweep.sh  -- C4 on the streaming kernel: exchange / ring depth / sync / segment variants, one line each
one() { python bench.py --workload c4 --eval-kernel 5 --no-cpu --no-extra --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print('%-36s kernel_us=%7.2f frac=%.3f seg_rows=%d ntiles=%d env=%s' % (r['kernel'], r['kernel_us'], r['frac'], c['tile_rows'], c['ntiles'], c['env']))
"; }
one
VA_E5_XDPP=0 one
VA_E5_NSLOT=4 one
VA_E5_NSLOT=8 one
VA_E5_SYNC=1 one
VA_E5_SYNC=2 one
VA_E5_SYNC=4 one
VA_E5_NSLOT=4 VA_E5_SYNC=1 one
VA_E5_NSLOT=4 VA_E5_SYNC=2 one
VA_E5_NSLOT=4 one --tile-rows 250
VA_E5_NSLOT=4 one --tile-rows 418
