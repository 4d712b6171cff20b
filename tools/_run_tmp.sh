mkdir -p gpurun_out/r3w
timeout -k 10 300 python tools/f3sweep.py > gpurun_out/r3w/f3sweep.txt 2>&1; cat gpurun_out/r3w/f3sweep.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3w/gpu.log 2>&1; tail -3 gpurun_out/r3w/gpu.log
