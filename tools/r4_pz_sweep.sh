#!/bin/bash
# persistent kernel: slice-size sweep with the stamps build
O=gpurun_out/r4c; mkdir -p $O
export VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_pzst.so
for rows in 0 12 16 24; do
  echo "== PZ_ROWS=$rows" >> $O/sweep.log
  PZ_ROWS=$rows timeout -k 10 120 python tools/persist_probe.py c1 c2 sh161 >> $O/sweep.log 2>&1 || { echo "rc=$?" >> $O/sweep.log; break; }
done
cat $O/sweep.log
