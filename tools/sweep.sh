#!/bin/bash
# usage: tools/sweep.sh <workload> <steps> "<tile rows list>" [eval_kernel] [gen]
#   gen: run the workload's right-hand side as a traced + generated module (bench.py --generated)
w=$1; steps=$2; ek=${4:-0}; gen=""
[ "$5" = "gen" ] && gen="--generated"
for T in $3; do
  python bench.py --workload $w --steps $steps --warmup 50 --no-cpu --no-extra --tile-rows $T --eval-kernel $ek $gen 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('$w ek=$ek %-44s T=%3d ntiles=%4d kernel_us=%9.2f frac=%.3f evals/s=%.3e' % (j['roofline']['kernel'], j['config']['tile_rows'], j['config']['ntiles'], j['roofline']['kernel_us'], j['roofline']['frac'], j['value']))
"
done
