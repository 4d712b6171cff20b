#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O; export TMPDIR=/tmp
for v in full 1 2 3; do
  if [ $v = full ]; then unset VARANNEAL_AMD_LIB; else export VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_fbabl$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/a$v -o nn -- python3 tools/nn_c5x.py 40 1 > $O/a$v.log 2>&1
  python3 - $O/a$v $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'nnet_fb' in r['Name']: print('abl %s  %-30s avg_us=%.1f' % (sys.argv[2], r['Name'][:30], float(r['AverageNs']) / 1e3))
PY
done
find $O -name "*.csv" -size +1M -delete
