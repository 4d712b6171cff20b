#!/bin/bash
O=gpurun_out/r4g; mkdir -p $O
for t in prio=1 prio=0 prio=2; do
  for rep in 1 2; do
    python bench.py --steps 3000 --warmup 300 --no-extra --no-cpu --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', 'kernel_us %.3f frac %.4f' % (d['roofline']['kernel_us'], d['roofline']['frac']))" >> $O/prio.txt
  done
done
cat $O/prio.txt
