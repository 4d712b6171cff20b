#!/bin/bash
# usage: tools/profiles.sh <tag>   -- collect a round's evidence into gpurun_out/profiles_<tag>/ (copy what is to be
# judged into profiles/ afterwards).  Counter passes run on their own, with --kernel-trace only (tools/pmc.sh, tools/pmc_mem.sh).
tag=$1; out=gpurun_out/profiles_$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py > $out/bench_c3.json 2> $out/bench_c3.err; cut -c1-300 $out/bench_c3.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_eval -- python3 bench.py --steps 500 --warmup 50 --no-cpu --no-extra > $out/trace_eval.log 2>&1
cp $out/trace_eval/*/*kernel_stats.csv $out/eval_c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c4 -- python3 bench.py --workload c4 --steps 200 --warmup 20 --no-cpu --no-extra > $out/bench_c4_traced.json 2> $out/trace_c4.log
cp $out/trace_c4/*/*kernel_stats.csv $out/eval_c4_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_ladder -- python3 bench.py --mode ladder --no-cpu > $out/ladder_c3.json 2> $out/trace_ladder.log
cp $out/trace_ladder/*/*kernel_stats.csv $out/ladder_c3_kernel_stats.csv
./tools/pmc.sh $tag > $out/pmc_c3.txt 2>&1; grep "k_eval" $out/pmc_c3.txt | cut -c1-400
python tools/pmc_traffic.py $tag lorenz96_D20_N1000_L7_B64_trapezoid > $out/pmc_traffic_c3.json; cat $out/pmc_traffic_c3.json
./tools/pmc.sh ${tag}c4 --workload c4 > $out/pmc_c4.txt 2>&1; grep "k_eval" $out/pmc_c4.txt | cut -c1-400
python tools/pmc_traffic.py ${tag}c4 lorenz96_D200_N5000_L80_B64_trapezoid > $out/pmc_traffic_c4.json; cat $out/pmc_traffic_c4.json
./tools/pmc_mem.sh ${tag}c4 --workload c4 > $out/pmc_mem_c4.txt 2>&1; grep "k_eval" $out/pmc_mem_c4.txt | cut -c1-300
cp profiles/pmc_traffic.json $out/pmc_traffic.json
for w in c3 c3x4 c3x16 c3x64; do ./tools/sweep.sh $w 300 "0" 0; done > $out/sweep_batch.txt 2>&1; cat $out/sweep_batch.txt
# state widths: built-in / traced + generated on the kernel the geometry picks / generated on the flat kernel
for w in c3 w100 c4 w500; do st=300; [ $w = c3 ] || st=40; ./tools/sweep.sh $w $st "0" 0; ./tools/sweep.sh $w $st "0" 3; ./tools/sweep.sh $w $st "0" 0 gen; done > $out/sweep_width.txt 2>&1; cat $out/sweep_width.txt
python bench.py --workload c4 --steps 200 --warmup 20 --no-cpu --no-extra > $out/bench_c4.json 2>/dev/null; cut -c1-200 $out/bench_c4.json
python bench.py --workload c2 --no-cpu --no-extra > $out/bench_c2.json 2>/dev/null
python tools/f3sweep.py > $out/f3_variants.txt 2>&1
rm -rf $out/trace_eval $out/trace_ladder $out/trace_c4
