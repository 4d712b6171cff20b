#!/bin/bash
# rocprofv3 kernel statistics of the c5x evaluation, fused (argument 1) or separate kernels (0)
F=${1:-1}; O=$PWD/gpurun_out/nnfb; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$F -- python3 $R/tools/nn_c5x.py 40 $F > $O/trace$F.log 2>&1
cd $R
f=$(ls $O/trace$F/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $O/kernel_stats_fused$F.csv; cut -d, -f1-4 $O/kernel_stats_fused$F.csv | sed -n 1,12p; else echo "no stats file"; tail -5 $O/trace$F.log; fi
rm -rf $O/trace$F
