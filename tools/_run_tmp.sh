mkdir -p gpurun_out/r3u
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3u/gpu.log 2>&1; tail -3 gpurun_out/r3u/gpu.log
grep -q failed gpurun_out/r3u/gpu.log && exit 1
timeout -k 10 200 ./tools/e4abl.sh > gpurun_out/r3u/abl.txt 2>&1; cat gpurun_out/r3u/abl.txt
timeout -k 10 200 ./tools/e4abl.sh --tune graph=0 > gpurun_out/r3u/abl_eager.txt 2>&1; tail -6 gpurun_out/r3u/abl_eager.txt
