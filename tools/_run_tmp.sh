mkdir -p gpurun_out/r3z
timeout -k 10 600 python -m pytest tests/test_gpu_codegen.py -m gpu -q -x -s -k "ring_of_units" 2>&1 | grep -E "us per complete|passed|failed" | cut -c1-300
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3z/gpu.log 2>&1; tail -3 gpurun_out/r3z/gpu.log
