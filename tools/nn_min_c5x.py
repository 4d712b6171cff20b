"""usage: python3 tools/nn_min_c5x.py [iterations]  -- a short L-BFGS minimisation at c5x (line-search evaluations: the
trial point and the direction enter the evaluation) with the fused kernel and with the separate kernels"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from varanneal_amd import _capi, twin

w = bench.NNET_WORKLOADS["c5x"]
s, M, B = np.array(w["structure"]), w["M"], w["B"]
din, dout, _ = twin.make_nnet_twin(s, M)
RM = 1.0 / 0.005 ** 2
RF0 = 1.0e-8 * RM * float(np.sum(s) - s[0]) / float(s[0] + s[-1])
g = [twin.nnet_initial_guess(s, M, b) for b in range(B)]
Pidx = g[0][2]
P = np.array([x[1] for x in g])
XP = np.array([np.append(x[0], x[1][Pidx]) for x in g])
opts = {'gtol': 1e-12, 'ftol': 1e-14, 'maxfun': 100000, 'maxiter': int(sys.argv[1]) if len(sys.argv) > 1 else 30}
with _capi.NnetProblem(B, s, din, dout, [np.arange(s[0]), np.arange(s[-1])], RM, RF0, P, Pidx) as pb:
    for fused in (1, 0, 1, 0):
        pb.tune(nnet_fused=fused)
        pb.minimize_lbfgs(XP, 1.1 ** 60, dict(opts, maxiter=3))
        t0 = time.perf_counter()
        r = pb.minimize_lbfgs(XP, 1.1 ** 60, opts)
        dt = time.perf_counter() - t0
        print("fused=%d: %.1f ms, nit %d..%d, nfev max %d -> %.1f us per L-BFGS cycle, A[0] = %.12e" % (
            fused, dt * 1e3, r["nit"].min(), r["nit"].max(), r["nfev"].max(), dt * 1e6 / r["nfev"].max(), r["A"][0]), flush=True)
