mkdir -p gpurun_out/r3k
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3k/gpu_all.log 2>&1; tail -8 gpurun_out/r3k/gpu_all.log
