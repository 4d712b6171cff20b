mkdir -p gpurun_out/r3l
timeout -k 10 600 python -m pytest tests/test_gpu_annealer.py tests/test_gpu_parity.py tests/test_gpu_codegen.py -m gpu -q > gpurun_out/r3l/gpu.log 2>&1; tail -4 gpurun_out/r3l/gpu.log
python bench.py > gpurun_out/r3l/bench_c3.json 2> gpurun_out/r3l/bench_c3.err; cut -c1-1500 gpurun_out/r3l/bench_c3.json
