"""usage: VARANNEAL_AMD_LIB=.../libvaranneal_amd_taill2.so python3 tools/tail_scope_probe.py
DIAGNOSTIC ONLY: the C3 evaluation with the tail's exchange (partial rows, arrival counter) at L2 scope -- correct only while
every tile of a seed runs on one XCD, which nothing guarantees -- to size what an XCD-local tail would buy.  Prints the time per
evaluation and whether A and the gradient still equal the product library's."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from varanneal_amd import _capi, twin
D, N, B = 20, 1000, 64
Y, Lidx, XP, P = bench.make_inputs(D, N, B, 0)
with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid") as pb:
    A, me, fe, g = pb.action_grad(XP, bench.RF_SCALE)
    ks = bench.event_timed(pb, bench.RF_SCALE, 2000)
    A2, _, _, g2 = pb.action_grad(XP, bench.RF_SCALE)
print("lib %s: %.2f us per evaluation; A sum %.15e, |g| sum %.15e; repeat identical: %s" % (
    os.path.basename(os.environ.get("VARANNEAL_AMD_LIB", "libvaranneal_amd.so")), ks * 1e6, A.sum(), np.abs(g).sum(),
    bool(np.array_equal(A, A2) and np.array_equal(g, g2))), flush=True)
