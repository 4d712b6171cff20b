"""NaKL (Hodgkin-Huxley type) neuron model as the reference's tutorial defines it
(examples/jupyter-tutorial/VarAnneal_tutorial.ipynb, "NaKL" section): D = 4 states
(V, m, h, n), 18 parameters, one external stimulus I(t).  Test input (a model definition
is user-side code, like the reference's example scripts)."""
import numpy as np


def x_inf(V, Vt, Vs):
    return 0.5 * (1.0 + np.tanh((V - Vt) / Vs))


def x_tau(V, Vt, Vs, t1, t2):
    return t1 + t2 * (1.0 - np.tanh((V - Vt) / Vs) ** 2)


def nakl(t, x, pstim):
    dxdt = np.zeros_like(x)
    p, Iext = pstim
    V, m, h, n = (x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    gNa, gK, gL = (p[0], p[1], p[2])
    ENa, EK, EL = (p[3], p[4], p[5])
    Vtm, Vsm, t1m, t2m = (p[6], p[7], p[8], p[9])
    Vth, Vsh, t1h, t2h = (p[10], p[11], p[12], p[13])
    Vtn, Vsn, t1n, t2n = (p[14], p[15], p[16], p[17])
    dxdt[:, 0] = gNa * m ** 3 * h * (ENa - V) + gK * n ** 4 * (EK - V) + gL * (EL - V) + Iext
    dxdt[:, 1] = (x_inf(V, Vtm, Vsm) - m) / x_tau(V, Vtm, Vsm, t1m, t2m)
    dxdt[:, 2] = (x_inf(V, Vth, Vsh) - h) / x_tau(V, Vth, Vsh, t1h, t2h)
    dxdt[:, 3] = (x_inf(V, Vtn, Vsn) - n) / x_tau(V, Vtn, Vsn, t1n, t2n)
    return dxdt


# parameter ranges of the tutorial (cell "Pb"), used for initial guesses and bounds
PB = [[60.0, 180.0], [10.0, 30.0], [0.15, 0.45], [47.5, 52.5], [-80.85, -73.15], [-56.7, -51.3],
      [-42.0, -38.0], [14.25, 15.75], [0.095, 0.105], [0.38, 0.42], [-63.0, -57.0],
      [-15.75, -14.25], [0.95, 1.05], [6.65, 7.35], [-57.75, -52.25], [28.5, 31.5],
      [0.95, 1.05], [4.75, 5.25]]
STATE_BOUNDS = [[-100.0, 100.0], [0.0, 1.0], [0.0, 1.0], [0.0, 1.0]]


def l96_damped(t, x, p):
    """a Lorenz-96 variant NOT in the built-in registry: two parameters (forcing, damping)
    and an explicit time dependence -- exercises the generic path with D = 12."""
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[1] * x + p[0] * (1.0 + 0.1 * np.sin(t))[:, None]


def l96_damped_tdp(t, x, p):
    """l96_damped written for time-dependent parameters: p has one row per time point
    (va_ode.py:170-188 hands f the rows' own parameter vectors)."""
    return (np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[:, 1:2] * x
            + p[:, 0:1] * (1.0 + 0.1 * np.sin(t))[:, None])
