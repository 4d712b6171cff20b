"""State and parameter estimation in the NaKL (Hodgkin-Huxley type) neuron on MI355X.

Follows the NaKL section of the reference's examples/jupyter-tutorial/VarAnneal_tutorial.ipynb:
4 states (V, m, h, n), 18 parameters, an injected current as external stimulus
(`f(t, x, (p, stim))`, va_ode.py:345-354), per-component RF0, box bounds on states and
parameters.  The model is an ordinary Python function; varanneal_amd.codegen traces it,
differentiates it and compiles f, J^T v and (df/dp)^T v for gfx950 on first use.  The
tutorial's voltage recording is not shipped with the reference: a twin experiment is
integrated here instead.

    python examples/NaKL/NaKL_anneal.py [--nbeta 40] [--N 501]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from varanneal_amd import va_ode  # noqa: E402


def nakl(t, x, pstim):
    dxdt = np.zeros_like(x)
    p, Iext = pstim
    V, m, h, n = (x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    gNa, gK, gL, ENa, EK, EL = (p[0], p[1], p[2], p[3], p[4], p[5])
    dxdt[:, 0] = gNa * m ** 3 * h * (ENa - V) + gK * n ** 4 * (EK - V) + gL * (EL - V) + Iext
    for i, (Vt, Vs, t1, t2) in enumerate([(p[6], p[7], p[8], p[9]), (p[10], p[11], p[12], p[13]),
                                          (p[14], p[15], p[16], p[17])]):
        th = np.tanh((V - Vt) / Vs)
        dxdt[:, 1 + i] = (0.5 * (1.0 + th) - x[:, 1 + i]) / (t1 + t2 * (1.0 - th ** 2))
    return dxdt


P_TRUE = np.array([120.0, 20.0, 0.3, 50.0, -77.0, -54.0, -40.0, 15.0, 0.1, 0.4, -60.0, -15.0, 1.0, 7.0,
                   -55.0, 30.0, 1.0, 5.0])
PB = [[0.5 * v, 1.5 * v] if v > 0 else [1.5 * v, 0.5 * v] for v in P_TRUE]
XB = [[-100.0, 100.0], [0.0, 1.0], [0.0, 1.0], [0.0, 1.0]]


def twin_data(N, dt, seed=7):
    rng = np.random.RandomState(seed)
    t = dt * np.arange(N)
    Iext = 15.0 + 10.0 * np.sin(0.3 * t) + 8.0 * np.sin(0.07 * t + 1.0)
    x = np.array([-65.0, 0.05, 0.6, 0.3])
    X = np.empty((N, 4))
    sub = 10
    for n in range(N):
        X[n] = x
        for k in range(sub):
            I = np.array([Iext[n] + (Iext[min(n + 1, N - 1)] - Iext[n]) * k / sub])
            f = lambda y: nakl(None, y[None, :], (P_TRUE, I))[0]
            h = dt / sub
            k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
            x = x + h * (k1 + 2 * k2 + 2 * k3 + k4) / 6.0
    return t, Iext, X[:, :1] + 1.0 * rng.randn(N, 1), X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nbeta", type=int, default=40)
    ap.add_argument("--N", type=int, default=501)
    ap.add_argument("--out", default=".")
    args = ap.parse_args()
    dt = 0.02
    t, Iext, Y, truth = twin_data(args.N, dt)

    anneal1 = va_ode.Annealer()
    anneal1.set_model(nakl, 4)
    anneal1.set_data(Y, stim=Iext, t=t)

    rng = np.random.RandomState(1)
    X0 = np.column_stack([(XB[i][1] - XB[i][0]) * rng.rand(args.N) + XB[i][0] for i in range(4)])
    P0 = np.array([(b[1] - b[0]) * rng.rand() + b[0] for b in PB])
    RM = 1.0
    RF0 = [1.0e-8, 1.0e-4, 1.0e-4, 1.0e-4]                     # per-component RF0 (tutorial)
    BFGS_options = {'gtol': 1.0e-8, 'ftol': 1.0e-8, 'maxfun': 1000000, 'maxiter': 1000000}
    tstart = time.time()
    anneal1.anneal(X0, P0, 1.5, np.arange(args.nbeta), RM, RF0, [0], list(range(18)), dt_model=None,
                   init_to_data=True, disc='SimpsonHermite', method='L-BFGS-B', bounds=XB + PB,
                   opt_args=BFGS_options, adolcID=0)
    print("\nAnnealing completed in %f s." % (time.time() - tstart))
    print("relative error of the estimated parameters:", np.round(anneal1.P / P_TRUE - 1.0, 3))
    anneal1.save_paths(os.path.join(args.out, "paths.npy"))
    anneal1.save_params(os.path.join(args.out, "params.npy"))
    anneal1.save_action_errors(os.path.join(args.out, "action_errors.npy"))


if __name__ == "__main__":
    main()
