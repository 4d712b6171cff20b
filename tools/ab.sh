#!/bin/bash
# usage: tools/ab.sh <reps> [bench args]  -- same-box A/B of the C3 evaluation between builds of the library:
# every varanneal_amd/libvaranneal_amd_*.so present (diagnostic builds made by hand, never the product; the
# stamps build excluded) against the current library.  Box-to-box spread of the absolute numbers is ~4 %:
# only differences inside one call mean anything.
reps=${1:-3}; shift
one() { python bench.py --steps 3000 --warmup 300 --no-cpu --no-extra "$@" 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('%-28s kernel_us=%.3f frac=%.4f' % ('$TAG', j['roofline']['kernel_us'], j['roofline']['frac']))"; }
for i in $(seq $reps); do
  for so in varanneal_amd/libvaranneal_amd_*.so; do
    case $so in *stamps*) continue;; esac
    TAG=$(basename $so .so | sed 's/libvaranneal_amd_//') VARANNEAL_AMD_LIB=$PWD/$so one "$@"
  done
  TAG=current one "$@"
done
