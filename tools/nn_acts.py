"""usage: python3 tools/nn_acts.py  -- the c5x evaluation with each built-in activation, fused kernel against the separate kernels"""
import os, sys
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
for act in ("sigmoid", "tanh", "relu", "softplus", "linear"):
    out = []
    with _capi.NnetProblem(B, s, din, dout, [np.arange(s[0]), np.arange(s[-1])], RM, RF0, P, Pidx, act=act) as pb:
        for fused in (1, 0):
            pb.tune(nnet_fused=fused)
            pb.action_grad(XP, 1.1 ** 100)
            out.append(bench.event_timed(pb, 1.1 ** 100, 40) * 1e6)
    print("%-9s fused %.1f us | separate %.1f us" % (act, out[0], out[1]), flush=True)
