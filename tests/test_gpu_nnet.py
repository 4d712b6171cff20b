"""GPU: the feed-forward-network action (csrc/va_nnet.hip, MFMA f64 products) through the
C-ABI against what the reference's va_nnet.Annealer produced (tests/golden/nnet.npz) and,
at sizes the goldens do not reach, against the NumPy oracle."""
import numpy as np
import pytest

import va_nnet_oracle as vno
from _util import load_npz_cases
from varanneal_amd import _capi, twin

pytestmark = pytest.mark.gpu
SINGLE = ["g6_twin_rf1", "g6_twin_rf1e6", "g6_twin_noinit_rf1e4", "g6_ragged_full_rm2", "g6_ragged_tanh",
          "g6_linear_wide", "g6_mnistlike_100_30_10"]


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("nnet.npz")


def _rm(c):
    return c["RM"] if np.ndim(c["RM"]) else float(c["RM"])


@pytest.mark.parametrize("name", SINGLE)
def test_single_eval_matches_reference(gold, name):
    c = gold[name]
    B = 3
    rng = np.random.RandomState(11)
    XP = c["XP"]
    XPb = np.stack([XP, XP + 0.01 * rng.randn(XP.size), XP])
    pr = _capi.NnetProblem(B, c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                           np.tile(c["P"], (B, 1)), c["Pidx"], act=str(c["act"]))
    A, me, fe, g = pr.action_grad(XPb, float(c["rf_scale"]))
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * abs(c["A"])
        assert abs(me[b] - c["me"]) <= 1e-12 * abs(c["A"]) and abs(fe[b] - c["fe"]) <= 1e-12 * abs(c["A"])
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    assert np.array_equal(g[0], g[2]) and A[1] != A[0]
    # the perturbed seed against the oracle
    pb = vno.NnetProblem(c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                         c["P"], c["Pidx"], act=str(c["act"]))
    A1, me1, fe1, g1 = pb.action_grad(XPb[1], float(c["rf_scale"]))
    assert abs(A[1] - A1) <= 1e-12 * abs(A1) and np.abs(g[1] - g1).max() <= 1e-10 * np.abs(g1).max()
    pr.close()


@pytest.mark.parametrize("structure,M,act,weights_only", [
    ([64, 64, 64, 64], 256, "sigmoid", False),      # tiles exactly full
    ([70, 33, 129, 5], 300, "tanh", True),          # ragged in every dimension, 2 example chunks
    ([784, 30, 10], 37, "sigmoid", False),          # tutorial MNIST shape (VarAnneal_tutorial.ipynb:3413-3415)
])
def test_larger_shapes_against_oracle(structure, M, act, weights_only):
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(0, structure[-1], 2)]
    dout = dout[:, Lidx[1]]
    B = 2
    X0, P0, XPs = [], [], []
    for b in range(B):
        x, p, Pidx = twin.nnet_initial_guess(structure, M, b, weights_only)
        p = p + 0.01 * np.random.RandomState(b).randn(p.size)          # non-zero biases
        X0.append(x); P0.append(p); XPs.append(np.append(x, p[Pidx]))
    pr = _capi.NnetProblem(B, structure, din, dout, Lidx, [2.0, 3.0], 0.7, np.array(P0), Pidx, act=act)
    A, me, fe, g = pr.action_grad(np.array(XPs), 1.5)
    for b in range(B):
        pb = vno.NnetProblem(structure, din, dout, Lidx, np.array([2.0, 3.0]), 0.7, P0[b], Pidx, act=act)
        A1, me1, fe1, g1 = pb.action_grad(XPs[b], 1.5)
        assert abs(A[b] - A1) <= 1e-12 * abs(A1) and abs(me[b] - me1) <= 1e-12 * abs(A1)
        assert np.abs(g[b] - g1).max() <= 1e-10 * np.abs(g1).max()
    pr.close()
