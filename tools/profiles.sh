#!/bin/bash
# usage: tools/profiles.sh <tag>   -- collect the round's evidence into gpurun_out/profiles_<tag>/
tag=$1; out=gpurun_out/profiles_$tag; mkdir -p $out; export TMPDIR=/tmp
python bench.py > $out/bench_c3.json 2> $out/bench_c3.err; cat $out/bench_c3.json | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_eval -- python3 bench.py --steps 500 --warmup 50 --no-cpu > $out/trace_eval.log 2>&1
cp $out/trace_eval/*/*kernel_stats.csv $out/eval_c3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_ladder -- python3 bench.py --mode ladder > $out/ladder_c3.json 2> $out/trace_ladder.log
cp $out/trace_ladder/*/*kernel_stats.csv $out/ladder_c3_kernel_stats.csv
./tools/pmc.sh $tag > $out/pmc_c3.txt 2>&1; cat $out/pmc_c3.txt | grep eval3 | cut -c1-300
for w in c3 c3x4 c3x16 c3x64; do ./tools/sweep.sh $w 300 "0" 0; done > $out/sweep_batch.txt 2>&1; cat $out/sweep_batch.txt
python bench.py --workload c4 --steps 100 --warmup 10 --no-cpu > $out/bench_c4.json 2>/dev/null; cut -c1-160 $out/bench_c4.json
python bench.py --workload c2 --no-cpu > $out/bench_c2.json 2>/dev/null
./tools/ablate.sh c3 0 > $out/ablation_c3.txt 2>&1; cat $out/ablation_c3.txt
python bench.py --workload c5 > $out/bench_c5.json 2>/dev/null; cut -c1-160 $out/bench_c5.json
python bench.py --workload c5x --steps 30 --warmup 5 > $out/bench_c5x.json 2>/dev/null; cut -c1-160 $out/bench_c5x.json
for w in c5 c5x; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -- python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu > $out/trace_$w.log 2>&1
  cp $out/trace_$w/*/*kernel_stats.csv $out/nnet_${w}_kernel_stats.csv
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_peak tools/mfma_f64_peak.hip 2>/dev/null && ./tools/mfma_f64_peak > $out/mfma_f64_peak.txt; cat $out/mfma_f64_peak.txt
python tools/timeline.py > $out/timeline_c3.txt 2>&1; tail -12 $out/timeline_c3.txt
rm -rf $out/trace_eval $out/trace_ladder $out/trace_c5 $out/trace_c5x
