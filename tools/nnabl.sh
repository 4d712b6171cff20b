#!/bin/bash
# usage: tools/nnabl.sh <outdir>  -- c5x per-kernel times of the measurement builds (VA_NN_ABLATE 1..4) and the full build
out=$1; mkdir -p $out
export TMPDIR=/tmp
for v in full 1 2 3 4; do
  if [ $v = full ]; then unset VARANNEAL_AMD_LIB; else export VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_nnabl$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/v$v -o nn -- python3 tools/nn_c5x.py 40 > $out/v$v.log 2>&1 || exit 1
  echo "== $v: $(grep c5x $out/v$v.log)"
  python3 - $out/v$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'nnet' in r['Name']: print('   %-45s calls=%s avg_us=%.1f' % (r['Name'][:45], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
