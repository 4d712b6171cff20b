"""CPU: the oracle (C restatement + NumPy twin) against the golden vectors
generated from the reference's own action code (oracle/gen_golden.py).

Tolerances: |A-A_ref|/|A_ref| <= 1e-12,
||g-g_ref||_inf/||g_ref||_inf <= 1e-10 (we hold 1e-11)."""
import numpy as np
import pytest

from _util import oracle_problem

RTOL_A = 1e-12
RTOL_G = 1e-11


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_golden_inventory(golden_single):
    assert len(golden_single) == 33
    assert sum("grad" in c for c in golden_single.values()) == 25


def test_survey_known_answers(golden_single):
    # SURVEY.md Appendix A step 5 check values (reference run by the surveyor)
    assert golden_single["g1_trapezoid_rf4e-06_itd1"]["A"] == pytest.approx(1.6147162902115759e-04, rel=1e-14)
    assert golden_single["g1_SimpsonHermite_rf4e-06_itd1"]["A"] == pytest.approx(1.4435953272311613e-04, rel=1e-14)
    assert golden_single["g1_euler_rf4e-06_itd1"]["A"] == pytest.approx(1.587495238992746e-04, rel=1e-14)
    c = golden_single["g1_trapezoid_rf4e-06_itd0"]
    assert c["A"] == pytest.approx(189.114067333729, rel=1e-13)
    assert c["grad"][0] == pytest.approx(5.360970321774511e-02, rel=1e-12)
    assert c["grad"][-1] == pytest.approx(4.554845115195793e-08, rel=1e-10)
    assert np.abs(c["grad"]).max() == pytest.approx(1.266893e-01, rel=1e-6)


def test_c_oracle_values_and_grads(golden_single):
    for name, c in golden_single.items():
        pb = oracle_problem(c)
        A, me, fe, g = pb.action_grad(c["XP"], c["rf_scale"])
        assert rel(A, c["A"]) <= RTOL_A, name
        assert abs(me - c["me"]) <= RTOL_A * max(abs(c["me"]), abs(c["A"])), name
        assert rel(fe, c["fe"]) <= RTOL_A, name
        if "grad" in c:
            err = np.abs(g - c["grad"]).max() / np.abs(c["grad"]).max()
            assert err <= RTOL_G, (name, err)


def test_numpy_twin_values(golden_single):
    for name, c in golden_single.items():
        pb = oracle_problem(c)
        A, me, fe = pb.numpy_action(c["XP"], c["rf_scale"])
        assert rel(A, c["A"]) <= 1e-14, name
        assert rel(fe, c["fe"]) <= 1e-14, name


def test_numpy_twin_complex_step_matches_c_adjoint(golden_single):
    # independent of the fixtures: complex-step through the twin on a fresh point
    c = golden_single["g3_vecRMRF_SimpsonHermite"]
    pb = oracle_problem(c)
    rng = np.random.RandomState(3)
    XP = c["XP"] + 0.1 * rng.randn(c["XP"].size)
    _, _, _, g = pb.action_grad(XP, 7.0)
    idx = rng.choice(XP.size, 60, replace=False)
    for i in list(idx) + [XP.size - 1]:
        z = XP.astype(complex); z[i] += 1e-30j
        gi = pb.numpy_action(z, 7.0)[0].imag / 1e-30
        assert abs(gi - g[i]) <= 1e-11 * np.abs(g).max()


def test_simpson_hermite_requires_odd_N(golden_single):
    import va_oracle
    c = golden_single["g1_SimpsonHermite_rf4e-06_itd1"]
    N = 160
    pb = va_oracle.Problem(20, N, c["Y"][:N], c["Lidx"], 0.025, 4.0, 4e-6, [8.0], [0],
                           disc="SimpsonHermite")
    with pytest.raises(ValueError):
        pb.action_grad(np.zeros(N * 20 + 1))


@pytest.mark.parametrize("name", ["g4_c1_trapezoid_N200", "g4_shipped_SH_N161"])
def test_oracle_rung_by_rung_from_the_references_start(golden_ladders, name):
    """The CPU oracle (restated L-BFGS-B + fused adjoint) started at the REFERENCE's own start point of every rung
    (tests/golden/ladder_paths.npz: the minimisers its anneal() + SciPy stored, va_ode.py:776) ends where the reference
    did on at least 27 of 30 rungs (A within 1e-3); rungs that part ways are long minimisations."""
    import va_oracle
    from _util import load_npz_cases
    c = golden_ladders[name]
    N, D = int(c["N"]), int(c["D"])
    ND = N * D
    paths = load_npz_cases("ladder_paths.npz")[name]["minpaths"]
    X0 = c["X0"].copy(); X0[:, c["Lidx"]] = c["Y"]
    XP0 = np.append(X0.ravel(), c["P0"])
    rf = float(c["alpha"]) ** c["beta"].astype(np.uint16)
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}
    off, same_nit = [], 0
    for k in range(len(rf)):
        start = paths[k - 1] if k else XP0
        pb = va_oracle.Problem(D, N, c["Y"], c["Lidx"], float(c["t"][1] - c["t"][0]), 4.0, 4e-6, start[ND:], [0], disc=str(c["disc"]))
        x, A, st, nit, nfev = pb.minimize_lbfgs(start, rf[k], opts)
        assert st == 0
        same_nit += int(nit == int(c["nit"][k]))
        if abs(A - c["A_array"][k]) > 1e-3 * c["A_array"][k]:
            off.append((k, A, float(c["A_array"][k]), nit, int(c["nit"][k])))
    print(name, "oracle vs reference, rung-local: off", off, "same nit on", same_nit)
    assert len(off) <= 3 and all(q[3] >= 50 and q[4] >= 50 for q in off), off
    assert same_nit >= 20
