import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from varanneal_amd import _capi, codegen, twin
from test_gpu_codegen import _ring_of_units
D, NP, N, B = 20, 3, 1000, 64
P = np.array([0.7, 0.9, 1.3])
t, Y, _, Lidx = twin.make_twin(D, N)
rng = np.random.RandomState(8)
XP = np.concatenate([0.8 * rng.randn(B, N * D), np.tile(P, (B, 1))], axis=1)
for tr in (48, 60, 72, 84, 96):
    m = codegen.module_for(_ring_of_units, D, NP, col_variant=lambda ne, gh: _capi.eval_plan(B, D, N, "trapezoid", ne, gh, tile_rows=tr))
    rid = _capi.load_rhs_module(m["so"])
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 0.3, np.tile(P, (B, 1)), [0, 1, 2], disc="trapezoid", rhs=rid, tile_rows=tr) as pr:
        pr.action_grad(XP, 2.0); pr.eval_timed(2.0, 100)
        us = min(pr.eval_timed(2.0, 500) for _ in range(3)) * 2
        print(tr, m["col_variant"], pr.info()["eval_kernel"], pr.info()["run_rows"], pr.info()["ntiles"], "%.2f us" % us, flush=True)
