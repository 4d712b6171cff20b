#!/bin/bash
# what the folded tail costs k_eval4 at C3: rocprofv3 averages with the tail in the kernel (default) and as a kernel of its own (tune fold=0)
O=$PWD/gpurun_out/tailcost; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for f in 1 0; do
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$f -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu --no-extra --tune fold=$f > $O/t$f.log 2>&1
  cd $R; g=$(ls $O/t$f/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "fold=$f"; if [ -n "$g" ]; then python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$g')))[:3]: print('  %-70s calls %6s  avg %9.1f ns' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])))"; else tail -3 $O/t$f.log; fi
  rm -rf $O/t$f
done
