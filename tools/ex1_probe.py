#!/usr/bin/env python3
"""The shipped example's problem (examples/Lorenz96_D20/Lorenz96_anneal.py: ONE seed, N = 161, Simpson-Hermite,
keep_paths) through the C-ABI binding directly, with the ladder length and the hipGraph knob on the command line:
the probe behind DESIGN.md's note on running that ladder under rocprofv3.

    python tools/ex1_probe.py [--nbeta 101] [--graph 1] [--paths 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from varanneal_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nbeta", type=int, default=101)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--paths", type=int, default=1)
args = ap.parse_args()
D, Lidx = 20, [0, 2, 4, 6, 8, 10, 14, 16]
data = np.load(os.path.join(ROOT, "tests", "golden", "l96_D20_dt0p025_N161_sm0p5_sec1_mem1.npy"))
t = data[:, 0]; Y = np.ascontiguousarray(data[:, 1:][:, Lidx]); N = len(t)
rng = np.random.RandomState(12345)
X0 = (20.0 * rng.rand(N * D) - 10.0).reshape(N, D); P0 = np.array([4.0 * rng.rand() + 6.0])
X0[:, Lidx] = Y
XP = np.append(X0.ravel(), P0)[None, :]
opts = {'gtol': 1.0e-8, 'ftol': 1.0e-8, 'maxfun': 1000000, 'maxiter': 1000000}
rf = 1.5 ** np.arange(args.nbeta)
with _capi.Problem(1, D, N, Y, Lidx, float(t[1] - t[0]), 4.0, 4e-6, P0[None, :], [0], disc="SimpsonHermite",
                   max_beta=args.nbeta, keep_paths=args.paths) as pb:
    pb.tune(graph=args.graph)
    t0 = time.time()
    r = pb.anneal(XP, rf, opts, want_paths=bool(args.paths))
    dt = time.time() - t0
    c = pb.counters()
print("probe ok: nbeta=%d graph=%d paths=%d  %.3f s  cycles=%d  A_final=%.6e  k=%.5f" % (
    args.nbeta, args.graph, args.paths, dt, c["cycles"], r["A"][0, -1], r["pest"][0, -1, 0]), flush=True)
