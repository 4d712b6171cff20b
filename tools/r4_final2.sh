#!/bin/bash
# end of round 4: counters on the final sources (traffic hash), the contract line at the defaults and at the driver's flags
O=gpurun_out/profiles_r04b; mkdir -p $O; export TMPDIR=/tmp
./tools/pmc.sh r04b > $O/pmc_c3.txt 2>&1; python tools/pmc_traffic.py r04b lorenz96_D20_N1000_L7_B64_trapezoid > $O/pmc_traffic_c3.json
./tools/pmc.sh r04bc4 --workload c4 > $O/pmc_c4.txt 2>&1; python tools/pmc_traffic.py r04bc4 lorenz96_D200_N5000_L80_B64_trapezoid > $O/pmc_traffic_c4.json
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3_driver_flags.json 2> $O/bench_c3_driver_flags.err; echo "bench driver rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eval -- python3 bench.py --steps 500 --warmup 50 --no-cpu --no-extra > $O/trace_eval.log 2>&1
cp $O/trace_eval/*/*kernel_stats.csv $O/eval_c3_kernel_stats.csv; rm -rf $O/trace_eval
cat $O/pmc_traffic.json | head -30
