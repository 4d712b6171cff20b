mkdir -p gpurun_out/r3x
timeout -k 10 600 python -m pytest tests/test_gpu_codegen.py -m gpu -q -x -k "bounded or nakl" > gpurun_out/r3x/gpu.log 2>&1; tail -30 gpurun_out/r3x/gpu.log
