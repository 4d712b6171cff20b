"""CPU: time-dependent parameters (P0 of shape (N_model, NP), va_ode.py:170-188) through the
shared tile phases (emulator) against what the reference produced (tests/golden/tdp.npz,
oracle/gen_golden.py:tdp_cases) and, for a generated right-hand side with only some of the
parameters estimated, against complex-step derivatives of the NumPy restatement."""
import numpy as np
import pytest

import va_oracle
from _util import load_npz_cases
from cpu_emul import emul
from models.nakl import l96_damped_tdp
from varanneal_amd import _capi, codegen, twin


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("tdp.npz")


SINGLE = ["g8_tdp_trapezoid_rf1e+00", "g8_tdp_trapezoid_rf3e+03", "g8_tdp_SimpsonHermite_rf1e+00",
          "g8_tdp_SimpsonHermite_rf3e+03"]


@pytest.mark.parametrize("name", SINGLE)
def test_oracle_and_emulator_match_reference(gold, name):
    c = gold[name]
    D, N = int(c["D"]), int(c["N_model"])
    XP, P0, Lidx = c["XP"], c["P0"], list(c["Lidx"])
    rf = float(c["rf_scale"])
    A0, me0, fe0 = va_oracle.numpy_action_generic(twin.l96, XP, D, N, c["Y"], Lidx, float(c["dt_model"]), 4.0,
                                                  4e-6 * rf, 1, [0], P0, str(c["disc"]))
    assert abs(A0 - c["A"]) <= 1e-12 * c["A"]
    desc, keep = _capi.make_desc(1, D, N, c["Y"], Lidx, float(c["dt_model"]), 4.0, 4e-6, P0[None], [0],
                                 disc=str(c["disc"]), p_time_dependent=True)
    A, me, fe, g = emul.action_grad(desc, 7, XP[None, :], rf)
    assert abs(A[0] - c["A"]) <= 1e-12 * c["A"] and abs(fe[0] - c["fe"]) <= 1e-12 * c["A"]
    assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    # away from the data (me != 0): complex step through the restatement
    rng = np.random.RandomState(3)
    XP2 = XP + 0.3 * rng.randn(XP.size)
    fun = lambda z: va_oracle.numpy_action_generic(twin.l96, z, D, N, c["Y"], Lidx, float(c["dt_model"]), 4.0,
                                                   4e-6 * rf, 1, [0], P0, str(c["disc"]))
    A2, me2, fe2, g2 = emul.action_grad(desc, 16, XP2[None, :], rf)
    assert abs(A2[0] - fun(XP2)[0]) <= 1e-12 * A2[0] and me2[0] > 0
    gc = va_oracle.complex_step_grad(fun, XP2)
    assert np.abs(g2[0] - gc).max() <= 1e-10 * np.abs(gc).max()


@pytest.mark.parametrize("disc,N", [("trapezoid", 30), ("SimpsonHermite", 31)])
def test_generated_rhs_partial_estimation(disc, N):
    """two parameters per time point, only the damping estimated (the reference's fe_gaussian
    supports this, va_ode.py:183-188; its anneal_step does not)."""
    D, NP = 12, 2
    m = codegen.module_for(l96_damped_tdp, D, NP, p_rows=True, compile=False)      # (header only: the emulator compiles it)
    rng = np.random.RandomState(5)
    t = 0.025 * np.arange(N)
    Lidx = [0, 2, 5, 7, 10]
    Y = rng.randn(N, 5)
    P = np.column_stack([8.0 + 0.5 * rng.randn(N), 1.0 + 0.1 * rng.randn(N)])
    XP = np.append(3.0 * rng.randn(N * D), P[:, 1])
    fun = lambda z: va_oracle.numpy_action_generic(l96_damped_tdp, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [1], P, disc,
                                                   t_model=t)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None], [1], disc=disc, rhs=1000, t_model=t,
                                 p_time_dependent=True)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    assert np.abs(g[0][N * D:]).max() > 0
