"""usage: python3 tools/lin_wide.py [D N B]  -- a dense-coupling model (varanneal_amd/twin.py dense_coupling_model) at a wide shape:
us per complete evaluation and the matrix-core rate of its two products (X C^T and S C: 4 N D^2 flop per seed)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from varanneal_amd import _capi, codegen, twin

D, N, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (200, 5000, 64)
m = codegen.module_for(twin.dense_coupling_model(D, D)[0], D, 2)
assert m["lin"] is not None
Lidx = list(range(0, D, 5))
rng = np.random.RandomState(0)
Y = rng.randn(N, len(Lidx))
P = np.tile(np.array([1.5, 0.3]), (B, 1))
XP = np.concatenate([rng.randn(B, N * D), P], axis=1)
for tr in (0,) + tuple(int(a) for a in sys.argv[4:]):
    with _capi.Problem(B, D, N, Y, Lidx, 0.02, 2.0, 0.7, P, [0, 1], disc="trapezoid", rhs=_capi.load_rhs_module(m["so"]), tile_rows=tr) as pr:
        info = pr.info()
        pr.action_grad(XP, 3.0)
        pr.eval_timed(3.0, 3)
        us = min(pr.eval_timed(3.0, 10) for _ in range(3)) / 10 * 1e3
    fl = 4.0 * N * D * D * B
    print("D=%d N=%d B=%d tile_rows=%d T=%s: %.1f us per evaluation, %.2f TFLOP/s in the two products (%.3f of 78.6), %.3f of 8 TB/s at 16 B per element"
          % (D, N, B, tr, info["tile_rows"], us, fl / us / 1e6, fl / us / 1e6 / 78.6, 16.0 * B * N * D / (us * 1e-6) / 8e12), flush=True)
