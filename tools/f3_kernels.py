"""usage: python3 tools/f3_kernels.py  -- the weight-array / merr_nskip variants of the C3 shape on each evaluation kernel"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from varanneal_amd import _capi, twin

def run(ek, tile_rows, **kw):
    D, B = 20, 64
    disc, N, nskip = kw.get("disc", "trapezoid"), kw.get("N", 1000), kw.get("nskip", 1)
    Y, Lidx, XP, P = bench.make_inputs(D, N, B, 0)
    RM, RF0 = 4.0, 4e-6
    if kw.get("rf_vec"):
        RF0 = np.resize(4e-6 * (1.0 + 0.1 * np.arange(D)), (N - 1, D))
    if nskip > 1:
        Y = Y[::nskip]
    if kw.get("rm_vec"):
        RM = np.resize(4.0 * (1.0 + 0.1 * np.arange(len(Lidx))), Y.shape)
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, RM, RF0, P, [0], disc=disc, merr_nskip=nskip, tile_rows=tile_rows, eval_kernel=ek) as pb:
        info = pb.info()
        pb.action_grad(XP, bench.RF_SCALE)
        ks = bench.event_timed(pb, bench.RF_SCALE, 500)
    return info, ks * 1e6

for kw in ({"rf_vec": True}, {"rm_vec": True}, {"nskip": 2, "N": 1001}):
    for ek, tr in ((4, 0), (4, 96), (3, 0), (3, 48), (3, 96), (1, 0), (5, 0), (5, 64)):
        try:
            info, us = run(ek, tr, **kw)
            print("%-28s asked kernel %d tile_rows %3d -> kernel %d K=%d T=%d tiles=%d  %7.2f us" % (
                sorted(kw.items()), ek, tr, info["eval_kernel"], info["run_rows"], info["tile_rows"], info["ntiles"], us), flush=True)
        except Exception as e:
            print("%-28s asked kernel %d tile_rows %3d -> %s" % (sorted(kw.items()), ek, tr, str(e)[:80]), flush=True)
