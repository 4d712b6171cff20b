"""GPU: time-dependent parameters (P0 of shape (N_model, NP), va_ode.py:170-188) through the
C-ABI (`p_time_dependent`): single evaluations against the reference (tests/golden/tdp.npz),
the device ladder through the Annealer against the reference's anneal(), a generated right-hand
side with part of the parameters estimated against complex-step derivatives."""
import numpy as np
import pytest

import va_oracle
from _util import load_npz_cases
from models.nakl import l96_damped_tdp
from varanneal_amd import _capi, codegen, twin, va_ode

pytestmark = pytest.mark.gpu
OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("tdp.npz")


@pytest.mark.parametrize("name", ["g8_tdp_trapezoid_rf1e+00", "g8_tdp_trapezoid_rf3e+03",
                                  "g8_tdp_SimpsonHermite_rf1e+00", "g8_tdp_SimpsonHermite_rf3e+03"])
def test_single_eval_matches_reference(gold, name):
    c = gold[name]
    D, N = int(c["D"]), int(c["N_model"])
    XP, P0, Lidx = c["XP"], c["P0"], list(c["Lidx"])
    rf = float(c["rf_scale"])
    rng = np.random.RandomState(2)
    XPb = np.stack([XP, XP + 0.3 * rng.randn(XP.size)])
    pr = _capi.Problem(2, D, N, c["Y"], Lidx, float(c["dt_model"]), 4.0, 4e-6, np.stack([P0, P0]), [0],
                       disc=str(c["disc"]), p_time_dependent=True)
    assert pr.info()["n_var"] == N * (D + 1)
    A, me, fe, g = pr.action_grad(XPb, rf)
    assert abs(A[0] - c["A"]) <= 1e-12 * c["A"] and abs(fe[0] - c["fe"]) <= 1e-12 * c["A"]
    assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    fun = lambda z: va_oracle.numpy_action_generic(twin.l96, z, D, N, c["Y"], Lidx, float(c["dt_model"]), 4.0,
                                                   4e-6 * rf, 1, [0], P0, str(c["disc"]))
    A1, me1, fe1 = fun(XPb[1])
    assert abs(A[1] - A1) <= 1e-12 * A1 and abs(me[1] - me1) <= 1e-12 * A1 and me1 > 0
    gc = va_oracle.complex_step_grad(fun, XPb[1])
    assert np.abs(g[1] - gc).max() <= 1e-10 * np.abs(gc).max()
    pr.close()


def test_device_ladder_matches_reference(gold, tmp_path):
    c = gold["g8_tdp_ladder_SH_N41"]
    N, D = int(c["N"]), int(c["D"])
    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(c["Y"], t=c["t"])
    X0, P0 = c["X0"].copy(), c["P0"].copy()
    a.anneal(X0, P0, float(c["alpha"]), c["beta"], 4.0, 4e-6, list(c["Lidx"]), [0], dt_model=None,
             init_to_data=True, disc="SimpsonHermite", method='L-BFGS-B', opt_args=OPTS, adolcID=0, verbose=False)
    assert list(a.nit_array[:7]) == list(c["nit"][:7])
    assert np.all(np.abs(a.A_array[:7] - c["A_array"][:7]) <= 1e-6 * c["A_array"][:7])
    # later rungs are trajectory-sensitive (rounding-level changes of the reduction order move them
    # by tens of percent, in either direction): no worse than the reference's minima by 10 %
    assert np.all(a.A_array <= 1.1 * c["A_array"]) and np.all(a.A_array >= 0.5 * c["A_array"])
    assert a.minpaths.shape == (len(c["beta"]), N * D + N)
    assert np.array_equal(P0[:, 0], a.minpaths[-1, N * D:])
    A, g = a.A_gradA_taped(a.minpaths[-1])
    assert abs(A - a.A_array[-1]) <= 1e-12 * A
    a.save_params(str(tmp_path / "p.npy"))
    assert np.load(str(tmp_path / "p.npy")).shape == (len(c["beta"]), N, 1)
    # stepwise == fused, and a batch of two seeds reproduces the single run in slot 1
    s = va_ode.Annealer(); s.set_model(twin.l96, D); s.set_data(c["Y"], t=c["t"])
    s.anneal(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), c["beta"][:6], 4.0, 4e-6, list(c["Lidx"]), [0],
             disc="SimpsonHermite", opt_args=OPTS, verbose=False, fused=False)
    assert np.array_equal(s.A_array, a.A_array[:6]) and np.array_equal(s.minpaths, a.minpaths[:6])
    b = va_ode.Annealer(); b.set_model(twin.l96, D); b.set_data(c["Y"], t=c["t"])
    Xb = np.stack([c["X0"] + 1.0, c["X0"]]); Pb = np.stack([c["P0"] + 0.5, c["P0"]])
    b.anneal(Xb, Pb, float(c["alpha"]), c["beta"][:6], 4.0, 4e-6, list(c["Lidx"]), [0], disc="SimpsonHermite",
             opt_args=OPTS, verbose=False)
    assert np.array_equal(b.A_array[1], a.A_array[:6]) and np.array_equal(b.minpaths[1], a.minpaths[:6])
    assert Pb.shape == (2, N, 1) and np.array_equal(Pb[1, :, 0], b.minpaths[1, -1, N * D:])
    for x in (a, s, b):
        x.close()


@pytest.mark.parametrize("disc,N", [("trapezoid", 64), ("SimpsonHermite", 65)])
def test_generated_rhs_partial_estimation(disc, N):
    D, NP = 12, 2
    m = codegen.module_for(l96_damped_tdp, D, NP, p_rows=True)
    rid = _capi.load_rhs_module(m["so"])
    rng = np.random.RandomState(5)
    t = 0.025 * np.arange(N)
    Lidx = [0, 2, 5, 7, 10]
    Y = rng.randn(N, 5)
    P = np.column_stack([8.0 + 0.5 * rng.randn(N), 1.0 + 0.1 * rng.randn(N)])
    XP = np.append(3.0 * rng.randn(N * D), P[:, 1])
    fun = lambda z: va_oracle.numpy_action_generic(l96_damped_tdp, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [1], P, disc,
                                                   t_model=t)
    g0 = va_oracle.complex_step_grad(fun, XP)
    pr = _capi.Problem(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None], [1], disc=disc, rhs=rid, t_model=t,
                       p_time_dependent=True)
    A, me, fe, g = pr.action_grad(XP[None, :], 1.0)
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    # the device minimiser moves the estimated column only
    r = pr.minimize_lbfgs(XP[None, :], 1.0, {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 300, 'maxiter': 300})
    assert r["A"][0] < A[0]
    pr.close()
    with pytest.raises(_capi.VaError):
        _capi.Problem(1, D, N if disc == "trapezoid" else N, Y, Lidx, 0.025, 4.0, 0.3, P[None], [1], disc="euler",
                      rhs=rid, t_model=t, p_time_dependent=True)
