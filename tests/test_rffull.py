"""Full model-error precision matrices (RF0 of shape (D, D) or (N_model-1, D, D), va_ode.py:211-217, 631-634):
the NumPy restatement and the flat tile phases (CPU emulator) against what the reference produced for its
Simpson-Hermite branch (tests/golden/rffull.npz, oracle/gen_golden.py:rffull_cases); the other discretisations --
whose upstream branch (va_ode.py:218-222) contracts RF[i] with the whole diff array and cannot run -- against
complex-step derivatives of the restatement's per-row contraction; the same through the C-ABI on the GPU."""
import numpy as np
import pytest

import va_oracle
from _util import load_npz_cases
from varanneal_amd import _capi, twin

NAMES = ["g10_rffull_SimpsonHermite_const", "g10_rffull_SimpsonHermite_time"]


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("rffull.npz")


def _rf(c):
    N, D = int(c["N_model"]), int(c["D"])
    return np.ascontiguousarray(np.resize(c["RF0"], (N - 1, D, D)))


def _other_disc_case(disc, seed=0):
    D, N, Lidx = 8, 26, [0, 2, 5]
    rng = np.random.RandomState(40 + seed)
    t, Y, _, _ = twin.make_twin(D, N, Lidx=Lidx)
    RF0 = np.array([0.5 * np.eye(D) + 0.1 * rng.randn(D, D) for _ in range(N - 1)])
    XP = np.append(10.0 * rng.rand(N * D) - 5.0, 7.5)
    fun = lambda z: va_oracle.numpy_action_generic(twin.l96, z, D, N, Y, Lidx, twin.DT, 1.5, RF0 * 3.0, 1, [0], XP[N * D:],
                                                   disc)
    return D, N, Lidx, Y, RF0, XP, fun


@pytest.mark.parametrize("name", NAMES)
def test_oracle_and_emulator_match_reference(gold, name):
    from cpu_emul import emul
    c = gold[name]
    D, N, Lidx = int(c["D"]), int(c["N_model"]), list(c["Lidx"])
    XP, RF0 = c["XP"], _rf(c)
    rf = float(c["alpha"]) ** int(c["beta"])
    A0, me0, fe0 = va_oracle.numpy_action_generic(twin.l96, XP, D, N, c["Y"], Lidx, float(c["dt_model"]), float(c["RM"]),
                                                  RF0 * rf, 1, [0], XP[N * D:], "SimpsonHermite")
    assert abs(A0 - c["A"]) <= 1e-12 * c["A"] and abs(fe0 - c["fe"]) <= 1e-12 * c["A"]
    desc, keep = _capi.make_desc(1, D, N, c["Y"], Lidx, float(c["dt_model"]), float(c["RM"]), RF0, XP[None, N * D:], [0],
                                 disc="SimpsonHermite")
    assert desc.rf_kind == 2
    for T in (8, 30):
        A, me, fe, g = emul.action_grad(desc, T, XP[None, :], rf)
        assert abs(A[0] - c["A"]) <= 1e-12 * c["A"] and abs(fe[0] - c["fe"]) <= 1e-12 * c["A"]
        assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()


@pytest.mark.parametrize("disc", ["trapezoid", "euler", "forwardmap"])
def test_emulator_matches_the_per_row_contraction(disc):
    from cpu_emul import emul
    D, N, Lidx, Y, RF0, XP, fun = _other_disc_case(disc)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, twin.DT, 1.5, RF0, XP[None, N * D:], [0], disc=disc)
    A, me, fe, g = emul.action_grad(desc, 7, XP[None, :], 3.0)
    A0, me0, fe0 = fun(XP)
    assert abs(A[0] - A0) <= 1e-12 * A0 and abs(fe[0] - fe0) <= 1e-12 * A0
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_reference(gold, name):
    c = gold[name]
    D, N, Lidx = int(c["D"]), int(c["N_model"]), list(c["Lidx"])
    XP, RF0 = c["XP"], _rf(c)
    rf = float(c["alpha"]) ** int(c["beta"])
    XPb = np.stack([XP, XP + 0.1, XP])
    pr = _capi.Problem(3, D, N, c["Y"], Lidx, float(c["dt_model"]), float(c["RM"]), RF0, np.tile(XP[N * D:], (3, 1)),
                       [0], disc="SimpsonHermite")
    assert pr.info()["eval_kernel"] == 1
    A, me, fe, g = pr.action_grad(XPb, rf)
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * c["A"] and abs(fe[b] - c["fe"]) <= 1e-12 * c["A"]
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    r = pr.minimize_lbfgs(XPb, rf, {'gtol': 1e-8, 'ftol': 1e-10, 'maxfun': 200, 'maxiter': 200})
    assert np.all(r["A"] < A)
    pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("disc", ["trapezoid", "euler", "forwardmap"])
def test_device_matches_the_per_row_contraction(disc):
    D, N, Lidx, Y, RF0, XP, fun = _other_disc_case(disc, seed=1)
    g0 = va_oracle.complex_step_grad(fun, XP)
    pr = _capi.Problem(2, D, N, Y, Lidx, twin.DT, 1.5, RF0, np.tile(XP[N * D:], (2, 1)), [0], disc=disc)
    A, me, fe, g = pr.action_grad(np.stack([XP, XP]), 3.0)
    A0, me0, fe0 = fun(XP)
    assert abs(A[1] - A0) <= 1e-12 * A0 and abs(fe[1] - fe0) <= 1e-12 * A0
    assert np.abs(g[1] - g0).max() <= 1e-10 * np.abs(g0).max()
    pr.close()


@pytest.mark.gpu
def test_annealer_accepts_matrix_rf(gold):
    """(D, D) RF0 through the drop-in: resized over time (va_ode.py:631-632), scaled up the ladder"""
    from varanneal_amd import va_ode
    c = gold["g10_rffull_SimpsonHermite_const"]
    D, N = int(c["D"]), int(c["N_model"])
    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(c["Y"], t=c["t"])
    X0 = c["XP"][:N * D].reshape(N, D).copy()
    a.anneal(X0, c["XP"][N * D:].copy(), 1.5, np.arange(3), float(c["RM"]), c["RF0"], list(c["Lidx"]), [0],
             init_to_data=False, disc="SimpsonHermite", opt_args={'gtol': 1e-8, 'ftol': 1e-8}, verbose=False)
    assert a.RF0.shape == (N - 1, D, D) and a.A_array[0] < 0.1 * c["A"]
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12) and np.all(a.exitflags == 0)
    a.close()
