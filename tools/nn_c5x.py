"""usage: python3 tools/nn_c5x.py [evals]  -- the c5x network action alone (bench.py extra.c5x), for rocprofv3 runs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.extra_nnet(0, "c5x", int(sys.argv[1]) if len(sys.argv) > 1 else 40, fused=(int(sys.argv[2]) if len(sys.argv) > 2 else None))
print("c5x us_per_eval=%.1f TFLOPs=%.2f frac=%.3f" % (r["us_per_eval_launch"], r["achieved_TFLOPs"], r["frac_of_f64_mfma_peak"]), flush=True)
