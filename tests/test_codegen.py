"""CPU: generated right-hand sides (varanneal_amd/codegen.py).  A user callable is traced,
differentiated and emitted as `struct RhsUser`; the generated code is (1) compiled into
the CPU emulator and run through the same flat tile phases the device module uses, and
checked against complex-step derivatives through a NumPy restatement of the reference's
action with the ORIGINAL Python callable; (2) cross-compiled with hipcc for gfx950."""
import os

import numpy as np
import pytest

import va_oracle
from cpu_emul import emul
from models.nakl import PB, l96_damped, nakl
from varanneal_amd import _capi, codegen


def _nakl_problem(N=41, seed=0):
    rng = np.random.RandomState(seed)
    D, NP = 4, 18
    t = 0.02 * np.arange(N)
    stim = 20.0 * np.sin(0.7 * t) + 5.0 * rng.randn(N)
    Y = (-60.0 + 30.0 * rng.rand(N, 1))
    X = np.column_stack([-70.0 + 60.0 * rng.rand(N), 0.2 + 0.6 * rng.rand(N, 3)])
    P = np.array([b[0] + (b[1] - b[0]) * rng.rand() for b in PB])
    RF0 = np.resize(np.array([1e-4, 1.0, 1.0, 1.0]), (N - 1, D))      # per-component RF (tutorial)
    return D, NP, N, t, stim, Y, X, P, RF0


@pytest.fixture(scope="module")
def nakl_module():
    return codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1)


def test_trace_reproduces_the_callable():
    ex, sy = codegen.trace(nakl, 4, 18, 1, 1)
    assert len(ex) == 4 and "tanh" in str(ex[1])
    codegen.check_against(nakl, ex, sy, 4, 18, 1, 1)

    def branching(t, x, p):
        return x if x[0, 0] > 0 else -x
    with pytest.raises(TypeError):
        codegen.trace(branching, 3, 1)


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_generated_nakl_matches_complex_step(nakl_module, disc):
    D, NP, N, t, stim, Y, X, P, RF0 = _nakl_problem()
    Pidx = list(range(18))
    XP = np.append(X.ravel(), P)
    rf = 1.5 ** 6
    fun = lambda z: va_oracle.numpy_action_generic(nakl, z, D, N, Y, [0], 0.02, 1.0, RF0 * rf, NP, Pidx, P,
                                                   disc, t_model=t, stim=stim)
    A0, me0, fe0 = fun(XP)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, [0], 0.02, 1.0, RF0, P[None, :], Pidx, disc=disc, rhs=1000,
                                 t_model=t, stim=stim)
    A, me, fe, g = emul.action_grad(desc, 16, XP[None, :], rf, user_header=nakl_module["header"])
    assert abs(A[0] - A0) <= 1e-12 * abs(A0) and abs(me[0] - me0) <= 1e-12 * abs(A0)
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    # partial estimation: only 3 of the 18 parameters in XP (va_ode.py:177-181)
    Pidx = [1, 7, 16]
    XPs = np.append(X.ravel(), P[Pidx])
    fun = lambda z: va_oracle.numpy_action_generic(nakl, z, D, N, Y, [0], 0.02, 1.0, RF0 * rf, NP, Pidx, P,
                                                   disc, t_model=t, stim=stim)
    desc, keep = _capi.make_desc(1, D, N, Y, [0], 0.02, 1.0, RF0, P[None, :], Pidx, disc=disc, rhs=1000,
                                 t_model=t, stim=stim)
    A, me, fe, g = emul.action_grad(desc, 16, XPs[None, :], rf, user_header=nakl_module["header"])
    gs = va_oracle.complex_step_grad(fun, XPs)
    assert abs(A[0] - fun(XPs)[0]) <= 1e-12 * abs(A0)
    assert np.abs(g[0] - gs).max() <= 1e-10 * np.abs(gs).max()


def test_generated_time_dependent_rhs():
    D, NP, N = 12, 2, 31
    m = codegen.module_for(l96_damped, D, NP)
    rng = np.random.RandomState(4)
    t = 0.025 * np.arange(N)
    Y = rng.randn(N, 5); Lidx = [0, 2, 5, 7, 10]
    P = np.array([8.0, 1.1])
    XP = np.append(3.0 * rng.randn(N * D), P)
    fun = lambda z: va_oracle.numpy_action_generic(l96_damped, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [0, 1], P,
                                                   "trapezoid", t_model=t)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None, :], [0, 1], rhs=1000, t_model=t)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()


def test_module_cross_compiles_for_gfx950(nakl_module):
    assert os.path.exists(nakl_module["so"]) and os.path.getsize(nakl_module["so"]) > 10000
    import ctypes as C
    L = C.CDLL(nakl_module["so"])               # loads without a GPU; exports the module ABI
    v = (C.c_int * 5)()
    L.va_user_rhs_info(v)
    assert list(v)[:3] == [18, 4, 1]
    assert hasattr(L, "va_user_launch_eval")


def _golden_nakl():
    from _util import load_npz_cases
    return load_npz_cases("nakl.npz")


@pytest.mark.parametrize("name", ["g5_nakl_SimpsonHermite_rf1e+00", "g5_nakl_SimpsonHermite_rf5e+01",
                                  "g5_nakl_trapezoid_rf1e+00", "g5_nakl_trapezoid_rf2e+03"])
def test_generated_nakl_matches_reference_golden(nakl_module, name):
    """A, measurement/model split and gradient the reference itself produced for the tutorial's
    NaKL model (oracle/gen_golden.py:nakl_cases) -- pins both the NumPy restatement for generic
    f and the generated code run through the emulator."""
    c = _golden_nakl()[name]
    D, N = int(c["D"]), int(c["N_model"])
    RF0 = np.resize(c["RF0"], (N - 1, D))
    XP = c["XP"]
    P = XP[N * D:]
    Pidx = list(range(18))
    rf = float(c["rf_scale"])
    A0, me0, fe0 = va_oracle.numpy_action_generic(nakl, XP, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]),
                                                  RF0 * rf, 18, Pidx, P, str(c["disc"]), t_model=c["t"],
                                                  stim=c["stim"])
    assert abs(A0 - c["A"]) <= 1e-12 * c["A"] and abs(me0 - c["me"]) <= 1e-12 * c["A"]
    desc, keep = _capi.make_desc(1, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]), RF0, P[None, :], Pidx,
                                 disc=str(c["disc"]), rhs=1000, t_model=c["t"], stim=c["stim"])
    A, me, fe, g = emul.action_grad(desc, 16, XP[None, :], rf, user_header=nakl_module["header"])
    assert abs(A[0] - c["A"]) <= 1e-12 * c["A"]
    assert abs(me[0] - c["me"]) <= 1e-12 * c["A"] and abs(fe[0] - c["fe"]) <= 1e-12 * c["A"]
    assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
