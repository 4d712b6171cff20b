#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O; export TMPDIR=/tmp
for f in 1 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f$f -o nn -- python3 tools/nn_c5x.py 40 $f > $O/f$f.log 2>&1
  echo "== fused=$f: $(grep c5x $O/f$f.log)"
  python3 - $O/f$f <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'nnet' in r['Name']: print('   %-45s calls=%s avg_us=%.1f' % (r['Name'][:45], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
find $O -name "*.csv" -size +1M -delete
