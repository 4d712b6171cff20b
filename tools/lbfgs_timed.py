"""usage: python tools/lbfgs_timed.py [c3|c4]  -- the two L-BFGS vector kernels (k_update, k_direction) alone, full histories
(va_lbfgs_timed), at the C3 shape (default) or at C4's (D = 200, N = 5000: histories of 10 GB, nothing cache-resident);
us per launch and the rate over the bytes each must move, two repetitions.  For same-box comparisons of builds
(VARANNEAL_AMD_LIB=...)."""
import sys, numpy as np
sys.path.insert(0, '.')
import bench
from varanneal_amd import _capi, twin
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
D, N, B, iters = (200, 5000, 64, 20) if which == "c4" else (20, 1000, 64, 300)
m = 10
Y, Lidx, XP, P = bench.make_inputs(D, N, B, 0)
with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", max_beta=2) as pb:
    ld = pb.info()["ld"]
    pb.anneal(XP, 1.5 ** np.arange(2), {'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': 12})
    pb.lbfgs_timed(max(iters // 6, 2))
    r = [pb.lbfgs_timed(iters) for _ in range(2)]
# bytes (bench.extra_ladder's accounting): update reads x, d, g, gt + 2 m history vectors, writes x, g, s, y;
# direction reads g + 2 m history vectors, writes d
bu, bd = 8.0 * B * ld * (4 + 2 * m + 4), 8.0 * B * ld * (1 + 2 * m + 1)
for u, d in r:
    u, d = u * 1e-3 / iters, d * 1e-3 / iters
    print("%s: k_update %.1f us = %.2f TB/s (%.0f MB)   k_direction %.1f us = %.2f TB/s (%.0f MB)"
          % (which, u * 1e6, bu / u / 1e12, bu / 1e6, d * 1e6, bd / d / 1e12, bd / 1e6), flush=True)
