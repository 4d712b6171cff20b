"""usage: python -m varanneal_amd._build --variant fbst -DVA_FB_STAMPS; VARANNEAL_AMD_LIB=.../libvaranneal_amd_fbst.so python3 tools/nn_fb_probe.py [fused value]
Phase times of one workgroup of k_nnet_fb at c5x (measurement build -DVA_FB_STAMPS, csrc/va_measure.h)."""
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
with _capi.NnetProblem(B, s, din, dout, [np.arange(s[0]), np.arange(s[-1])], RM, RF0, P, Pidx) as pb:
    pb.tune(nnet_fused=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for _ in range(3):
        pb.action_grad(XP, 1.1 ** 100)
    ks = bench.event_timed(pb, 1.1 ** 100, 20)
    t = pb.debug_read_persist(8) / 100.0          # 100 MHz ticks -> us
names = ["input image", "product 1", "barrier", "epilogue A", "barrier", "product 2", "epilogue B", "kernel"]
print("evaluation %.1f us;  one workgroup of k_nnet_fb, us summed over the layers:" % (ks * 1e6))
print("  " + " | ".join("%s %.1f" % (n, v) for n, v in zip(names, t)))
