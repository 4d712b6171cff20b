"""usage: python tools/lbfgs_timed.py  -- the two L-BFGS vector kernels (k_update, k_direction) alone, full histories, at C3
(va_lbfgs_timed); us per launch, two repetitions.  For same-box comparisons of builds (VARANNEAL_AMD_LIB=...)."""
import sys, numpy as np
sys.path.insert(0, '.')
import bench
from varanneal_amd import _capi, twin
D, N, B = 20, 1000, 64
Y, Lidx, XP, P = bench.make_inputs(D, N, B, 0)
with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", max_beta=2) as pb:
    pb.anneal(XP, 1.5 ** np.arange(2), {'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': 30})
    pb.lbfgs_timed(50)
    r = [pb.lbfgs_timed(300) for _ in range(2)]
print(" ".join("upd %.2f dir %.2f" % (u * 1e3 / 300, d * 1e3 / 300) for u, d in r))
