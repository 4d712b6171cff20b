#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_nnet.py tests/test_nnet_extra.py -x -q -m gpu > $O/nnet_tests.log 2>&1; echo "tests rc=$?" ; tail -5 $O/nnet_tests.log
python bench.py --workload c5x --steps 40 --warmup 10 --no-cpu > $O/c5x_fused.json 2>$O/c5x_fused.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4h/c5x_fused.json").read().strip().splitlines()[-1])
print("c5x fused: %.1f us  %.2f TFLOP/s  frac %.3f" % (d["roofline"]["kernel_us"], d["roofline"]["achieved"], d["roofline"]["frac"]))
PY
