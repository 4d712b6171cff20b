#!/usr/bin/env python3
"""The persistent per-seed ladder kernel (csrc/va_persist.h) against the three-launch cycle on the same problems:
per-rung (A, nit, nfev, status) side by side and the time of each.

    python tools/persist_probe.py [case ...]     cases: c1 c2 sh161 b4 euler fwd rfvec nskip2 sh101
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from varanneal_amd import _capi, twin  # noqa: E402

OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


def problem(D, N, B, disc="trapezoid", nbeta=30, rf_vec=False, nskip=1, shipped=False):
    if shipped:
        Lidx = [0, 2, 4, 6, 8, 10, 14, 16]
        data = np.load(os.path.join(ROOT, "tests", "golden", "l96_D20_dt0p025_N161_sm0p5_sec1_mem1.npy"))
        Y = np.ascontiguousarray(data[:, 1:][:, Lidx]); N = len(Y); dt = float(data[1, 0] - data[0, 0])
        rng = np.random.RandomState(12345)
        X0 = (20.0 * rng.rand(N * D) - 10.0).reshape(N, D); P0 = np.array([4.0 * rng.rand() + 6.0])
        X0[:, Lidx] = Y
        XP = np.append(X0.ravel(), P0)[None, :]; P = P0[None, :]
    else:
        t, Y, _, Lidx = twin.make_twin(D, N)
        dt = twin.DT
        XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
        for b in range(B):
            X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
            XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    RF0 = 4e-6
    if rf_vec:
        RF0 = np.resize(4e-6 * (1.0 + 0.1 * np.arange(D)), (N - 1, D))
    if nskip > 1:
        Y = Y[::nskip]
    return dict(B=XP.shape[0], D=D, N=N, Y=Y, Lidx=Lidx, dt=dt, RF0=RF0, P=P, XP=XP, disc=disc, nskip=nskip, nbeta=nbeta)


CASES = {
    "c1": lambda: problem(20, 200, 1),
    "c2": lambda: problem(20, 1000, 1),
    "sh161": lambda: problem(20, 161, 1, disc="SimpsonHermite", shipped=True),
    "sh101": lambda: problem(20, 161, 1, disc="SimpsonHermite", shipped=True, nbeta=101),
    "b4": lambda: problem(20, 200, 4),
    "euler": lambda: problem(20, 200, 2, disc="euler"),
    "fwd": lambda: problem(20, 200, 1, disc="forwardmap", nbeta=8),
    "rfvec": lambda: problem(20, 200, 1, rf_vec=True),
    "nskip2": lambda: problem(20, 201, 1, nskip=2),
    "d10": lambda: problem(10, 300, 2),
}


ROWS = int(os.environ.get("PZ_ROWS", "0"))


def run(q, persist):
    rf = 1.5 ** np.arange(q["nbeta"])
    with _capi.Problem(q["B"], q["D"], q["N"], q["Y"], q["Lidx"], q["dt"], 4.0, q["RF0"], q["P"], [0], disc=q["disc"],
                       merr_nskip=q["nskip"], max_beta=q["nbeta"], keep_paths=1) as pb:
        pb.tune(persist=persist)
        if persist and ROWS:
            pb.tune(persist_rows=ROWS)
        geo = pb.persistent()
        pb.anneal(q["XP"].copy(), rf[:2], OPTS, want_paths=True)                 # warm-up
        c0 = pb.counters()["cycles"]
        t0 = time.time()
        r = pb.anneal(q["XP"].copy(), rf, OPTS, want_paths=True)
        dt = time.time() - t0
        cyc = pb.counters()["cycles"] - c0
        if persist and "_pzs" in os.environ.get("VARANNEAL_AMD_LIB", ""):
            st = pb.debug_read_persist(14)
            names = ["trial", "tile phases", "wave sums", "publish+dots", "gather ev", "ls_step|gather dots", "dot sums+params", "coeffs", "update+direction", "-", "-", "-", "loop top"]
            n = max(st[13], 1.0)
            print("   stamps (workgroup 0 of seed 0, us per cycle over %d cycles): " % n +
                  "  ".join("%s %.2f" % (names[i], st[i] * 0.01 / n) for i in range(13) if names[i] != "-") +
                  "  | total %.2f | shader clock %.0f MHz" % ((st[:9].sum() + st[11:13].sum()) * 0.01 / n, st[9] / max(st[10], 1.0) * 100.0), flush=True)
    return r, dt, cyc, geo


for name in (sys.argv[1:] or list(CASES)):
    q = CASES[name]()
    r0, t0, c0, _ = run(q, 0)
    r1, t1, c1, geo = run(q, 1)
    same = int(np.sum((r0["nit"] == r1["nit"]) & (r0["nfev"] == r1["nfev"]) & (r0["status"] == r1["status"])))
    relA = np.abs(r1["A"] - r0["A"]) / np.abs(r0["A"])
    first_bad = np.argmax(~((r0["nit"] == r1["nit"]).all(axis=0))) if same != r0["nit"].size else -1
    dx = np.abs(r1["x"] - r0["x"]).max()
    dmp = np.abs(r1["minpaths"] - r0["minpaths"]).max()
    print("%-7s geo=%s  3-launch %.4f s %d cyc (%.1f us/cyc) | persistent %.4f s %d cyc (%.1f us/cyc) | rungs equal (nit,nfev,status) %d/%d, "
          "first differing rung %d, max rel dA %.2e (rung0 %.2e), final A %.6e vs %.6e, k %.5f vs %.5f, max|dx| %.2e, max|dminpaths| %.2e" % (
              name, geo, t0, c0, t0 * 1e6 / max(c0, 1), t1, c1, t1 * 1e6 / max(c1, 1), same, r0["nit"].size, first_bad,
              relA.max(), relA[:, 0].max(), r0["A"][0, -1], r1["A"][0, -1], r0["pest"][0, -1, 0], r1["pest"][0, -1, 0], dx, dmp), flush=True)
