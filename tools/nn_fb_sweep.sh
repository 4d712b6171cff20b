#!/bin/bash
# c5x evaluation: separate kernels (0), fused (1), fused with the first workgroups' starts spread over t us (2 + t)
# (the phase probe needs the measurement build: python -m varanneal_amd._build --variant fbst -DVA_FB_STAMPS)
O=gpurun_out/nnfb; mkdir -p $O
python -m pytest tests/test_gpu_nnet.py -x -q -k fused 2>&1 | tail -3
for f in 0 1 12 22 32 42; do
  echo "fused=$f: $(python tools/nn_c5x.py 40 $f)"
done | tee $O/sweep.txt
export VARANNEAL_AMD_LIB=$PWD/varanneal_amd/libvaranneal_amd_fbst.so
(python tools/nn_fb_probe.py 1 && python tools/nn_fb_probe.py 27) | tee $O/probe.txt
