#!/bin/bash
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err || exit 11
python bench.py --steps 2000 --warmup 200 --no-extra --no-cpu > $O/bench_2000.json 2> $O/bench_2000.err || exit 12
python bench.py --workload mnist_shape > $O/mnist_shape.json 2> $O/mnist_shape.err || exit 13
timeout -k 10 300 python -m pytest tests/test_gpu_nnet.py -q -m gpu -k mnist > $O/mnist_test.log 2>&1 || exit 14
tail -2 $O/mnist_test.log
