"""CPU: the network-action oracle (oracle/va_nnet_oracle.py) against what the reference's
va_nnet.Annealer produced (tests/golden/nnet.npz, oracle/gen_golden_nnet.py)."""
import numpy as np
import pytest

import va_nnet_oracle as vno
from _util import load_npz_cases
from varanneal_amd import twin


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("nnet.npz")


def problem_for(c):
    RM = c["RM"] if np.ndim(c["RM"]) else float(c["RM"])
    return vno.NnetProblem(c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], RM, float(c["RF0"]),
                           c["P"], c["Pidx"], act=str(c["act"]))


SINGLE = ["g6_twin_rf1", "g6_twin_rf1e6", "g6_twin_noinit_rf1e4", "g6_ragged_full_rm2", "g6_ragged_tanh",
          "g6_linear_wide", "g6_mnistlike_100_30_10"]


@pytest.mark.parametrize("name", SINGLE)
def test_single_eval_matches_reference(gold, name):
    c = gold[name]
    pb = problem_for(c)
    A, me, fe, g = pb.action_grad(c["XP"], float(c["rf_scale"]))
    assert abs(A - c["A"]) <= 1e-12 * abs(c["A"])
    assert abs(me - c["me"]) <= 1e-12 * abs(c["A"]) and abs(fe - c["fe"]) <= 1e-12 * abs(c["A"])
    assert np.abs(g - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    A2, me2, fe2 = pb.action(c["XP"], float(c["rf_scale"]))
    assert abs(A2 - A) <= 1e-14 * abs(A)


def test_mnist_shape_single_eval_matches_reference():
    """the tutorial's network in full, 784-30-10 with M = 2 (VarAnneal_tutorial.ipynb:3413-3415; synthetic data of that
    shape), value and complex-step gradient through the reference's own A (tests/golden/nnet_mnist.npz)"""
    c = load_npz_cases("nnet_mnist.npz")["g6_mnist_784_30_10"]
    assert list(c["structure"]) == [784, 30, 10] and int(c["M"]) == 2 and c["XP"].size == 2 * 824 + 784 * 30 + 30 * 10
    A, me, fe, g = problem_for(c).action_grad(c["XP"], float(c["rf_scale"]))
    assert abs(A - c["A"]) <= 1e-12 * abs(c["A"]) and abs(fe - c["fe"]) <= 1e-12 * abs(c["A"])
    assert np.abs(g - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()


def test_surveyors_probe_value_shape(gold):
    """SURVEY.md 8(c) G5 quotes A = fe, me = 0 at RF = RF0 = 0.0038 for the twin's seeded
    start; our seeds differ (RandomState(1000+i)), the structure of the result must not."""
    c = gold["g6_twin_rf1"]
    assert c["me"] == 0.0 and c["A"] == c["fe"] and abs(c["RF0"] - 0.0038) < 1e-12
    assert c["XP"].size == 2300 and c["Pidx"].size == 1900 and c["P"].size == 2090


def test_loop_form_equals_vectorised(gold):
    c = gold["g6_ragged_full_rm2"]
    pb = problem_for(c)
    fe = pb.reference_loop_action(c["XP"], 1.0, twin.sigmoid)
    assert abs(fe - c["fe"]) <= 1e-13 * c["fe"]


def test_ladder_restatement_follows_reference(gold):
    c = gold["g7_twin_ladder"]
    Lidx = [np.arange(10), np.arange(10)]
    pb = vno.NnetProblem(c["structure"], c["din"], c["dout"], Lidx, float(c["RM"]), float(c["RF0"]), c["P0"].copy(),
                         c["Pidx"])
    r = pb.scipy_ladder(c["X0"], float(c["alpha"]), c["beta"],
                        {'gtol': 1e-12, 'ftol': 1e-12, 'maxfun': 1000000, 'maxiter': 1000000})
    # same optimiser, values that agree to ~1e-15: the first rungs follow the reference step for
    # step; the middle of this ladder is chaotic (different local minima are visited, SURVEY.md
    # 7.3-4) and both runs meet again at the top
    assert list(r["nit"][:2]) == list(c["nit"][:2])
    assert np.all(np.abs(r["A"][:2] - c["A_array"][:2]) <= 1e-8 * c["A_array"][:2])
    assert np.all(np.abs(r["A"][:4] - c["A_array"][:4]) <= 1e-4 * c["A_array"][:4])
    assert abs(r["A"][-1] - c["A_array"][-1]) <= 1e-4 * c["A_array"][-1]


def test_arbiter_agrees_with_scipy_from_the_same_start(gold):
    """The arbiter of the network ladders -- this oracle under the C oracle's own L-BFGS
    (vno.NnetProblem.minimize_lbfgs -> va_oracle.c: vao_lbfgs_generic) -- against SciPy's L-BFGS-B
    around the same function: identical iteration / evaluation counts on short runs, the same
    minimum on converged ones."""
    import scipy.optimize as opt
    c = gold["g7_twin_ladder"]
    Lidx = [np.arange(10), np.arange(10)]
    pb = vno.NnetProblem(c["structure"], c["din"], c["dout"], Lidx, float(c["RM"]), float(c["RF0"]), c["P0"].copy(),
                         c["Pidx"])
    X0 = c["X0"].copy()
    XP0 = np.append(X0.ravel(), c["P0"][c["Pidx"]])
    for rf, maxiter in ((1.0, 3), (1.1 ** 30, 6)):
        o = {'gtol': 1e-12, 'ftol': 1e-12, 'maxfun': 100000, 'maxiter': maxiter}
        x, A, st, nit, nfev = pb.minimize_lbfgs(XP0, rf, o)
        rs = opt.minimize(lambda z: (lambda r: (r[0], r[3]))(pb.action_grad(z, rf)), XP0, method='L-BFGS-B', jac=True,
                          options=o)
        # (a 10-evaluation line search in iteration 2 amplifies last-bit differences to ~1e-7 by iteration 3, ~1e-5 by iteration 6)
        assert (nit, nfev) == (rs.nit, rs.nfev) and abs(A - rs.fun) <= 1e-3 * rs.fun, (rf, nit, rs.nit, A, rs.fun)
    o = {'gtol': 1e-12, 'ftol': 1e-12, 'maxfun': 100000, 'maxiter': 100000}
    x, A, st, nit, nfev = pb.minimize_lbfgs(XP0, 1.0, o)
    rs = opt.minimize(lambda z: (lambda r: (r[0], r[3]))(pb.action_grad(z, 1.0)), XP0, method='L-BFGS-B', jac=True, options=o)
    assert st == 0 and rs.status == 0 and abs(A - rs.fun) <= 1e-3 * rs.fun
