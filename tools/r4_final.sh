#!/bin/bash
# round 4: the few-seed regime -- shipped example and C1 / C2 ladders, plain and under rocprofv3
O=gpurun_out/r4j; mkdir -p $O; export TMPDIR=/tmp
python examples/Lorenz96_D20/Lorenz96_anneal.py --out $O > $O/example_plain.log 2>&1; grep "completed\|forcing" $O/example_plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ex -o ex -- python3 examples/Lorenz96_D20/Lorenz96_anneal.py --out $O > $O/example_rocprof.log 2>&1; echo "example under rocprofv3 rc=$?"; grep "completed" $O/example_rocprof.log
cp $O/ex/*kernel_stats.csv $O/example_kernel_stats.csv 2>/dev/null || cp $O/ex/*/*kernel_stats.csv $O/example_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o c2 -- python3 bench.py --workload c2 --mode ladder --no-cpu > $O/ladder_c2.json 2> $O/ladder_c2.err; echo "c2 ladder under rocprofv3 rc=$?"
cp $O/c2/*kernel_stats.csv $O/ladder_c2_kernel_stats.csv 2>/dev/null || cp $O/c2/*/*kernel_stats.csv $O/ladder_c2_kernel_stats.csv
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
rm -rf $O/ex $O/c2 $O/*.npy
head -4 $O/example_kernel_stats.csv | cut -c1-200; head -4 $O/ladder_c2_kernel_stats.csv | cut -c1-200; cat $O/ladder_c2.json | cut -c1-300
