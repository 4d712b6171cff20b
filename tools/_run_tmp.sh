mkdir -p gpurun_out/r3y
timeout -k 10 300 python -m pytest tests/test_gpu_codegen.py -m gpu -q -x -s -k "nakl_bounded_ladder_on_the_device" 2>&1 | grep -E "bounded NaKL|passed|failed" | cut -c1-600
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3y/gpu.log 2>&1; tail -3 gpurun_out/r3y/gpu.log
