"""Full measurement precision matrices (RM of shape (L, L) or (N_data, L, L), va_ode.py:149-152,
617-618): the NumPy restatement and the flat tile phases (CPU emulator) against what the
reference produced (tests/golden/rmfull.npz, oracle/gen_golden.py:rmfull_cases); the same through
the C-ABI on the GPU."""
import numpy as np
import pytest

import va_oracle
from _util import load_npz_cases
from varanneal_amd import _capi, twin

NAMES = ["g9_rmfull_trapezoid_const", "g9_rmfull_SimpsonHermite_time", "g9_rmfull_euler_time"]


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("rmfull.npz")


def _rm(c):
    N, L = int(c["N_model"]), len(c["Lidx"])
    return np.resize(c["RM"], (N, L, L)) if c["RM"].ndim == 2 else c["RM"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_and_emulator_match_reference(gold, name):
    from cpu_emul import emul
    c = gold[name]
    D, N, Lidx = int(c["D"]), int(c["N_model"]), list(c["Lidx"])
    XP, RM = c["XP"], _rm(c)
    A0, me0, fe0 = va_oracle.numpy_action_generic(twin.l96, XP, D, N, c["Y"], Lidx, float(c["dt_model"]), RM,
                                                  float(c["RF0"]), 1, [0], XP[N * D:], str(c["disc"]))
    assert abs(A0 - c["A"]) <= 1e-12 * c["A"] and abs(me0 - c["me"]) <= 1e-12 * c["A"]
    desc, keep = _capi.make_desc(1, D, N, c["Y"], Lidx, float(c["dt_model"]), RM, float(c["RF0"]), XP[None, N * D:],
                                 [0], disc=str(c["disc"]))
    assert desc.rm_kind == 2
    A, me, fe, g = emul.action_grad(desc, 7, XP[None, :], 1.0)
    assert abs(A[0] - c["A"]) <= 1e-12 * c["A"] and abs(me[0] - c["me"]) <= 1e-12 * c["A"]
    assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_reference(gold, name):
    c = gold[name]
    D, N, Lidx = int(c["D"]), int(c["N_model"]), list(c["Lidx"])
    XP, RM = c["XP"], _rm(c)
    XPb = np.stack([XP, XP + 0.1, XP])
    pr = _capi.Problem(3, D, N, c["Y"], Lidx, float(c["dt_model"]), RM, float(c["RF0"]), np.tile(XP[N * D:], (3, 1)),
                       [0], disc=str(c["disc"]))
    A, me, fe, g = pr.action_grad(XPb, 1.0)
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * c["A"] and abs(me[b] - c["me"]) <= 1e-12 * c["A"]
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    r = pr.minimize_lbfgs(XPb, 1.0, {'gtol': 1e-8, 'ftol': 1e-10, 'maxfun': 200, 'maxiter': 200})
    assert np.all(r["A"] < A)
    pr.close()


@pytest.mark.gpu
def test_annealer_accepts_matrix_rm(gold):
    from varanneal_amd import va_ode
    c = gold["g9_rmfull_trapezoid_const"]
    D, N = int(c["D"]), int(c["N_model"])
    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(c["Y"], t=c["t"])
    X0 = c["XP"][:N * D].reshape(N, D).copy()
    a.anneal(X0, c["XP"][N * D:].copy(), 1.5, np.arange(3), c["RM"], float(c["RF0"]), list(c["Lidx"]), [0],
             init_to_data=False, disc="trapezoid", opt_args={'gtol': 1e-8, 'ftol': 1e-8}, verbose=False)
    assert a.RM.shape == (N, 4, 4) and a.me_array[0] < 0.1 * c["me"]
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12) and np.all(a.exitflags == 0)
    a.close()
