"""CPU: the oracle's L-BFGS (published L-BFGS-B 3.0 unconstrained path +
MINPACK-2 dcsrch) pinned behaviourally against scipy.optimize.minimize --
the reference's actual minimiser (_autodiffmin.py:85-86) -- and the oracle's
ladder against the golden ladders the reference's anneal() produced.

Trajectory-level parity is chaotic: short runs must agree
step for step; long runs must land in the same basin within loose tolerance."""
import numpy as np
import scipy.optimize as opt

import va_oracle

OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


def _c1(golden_ladders, name="g4_c1_trapezoid_N200"):
    c = golden_ladders[name]
    N, D = int(c["N"]), int(c["D"])
    X0 = c["X0"].copy()
    X0[:, c["Lidx"]] = c["Y"]
    XP0 = np.append(X0.flatten(), c["P0"])
    pb = va_oracle.Problem(D, N, c["Y"], c["Lidx"], float(c["t"][1] - c["t"][0]), 4.0, 4e-6,
                           c["P0"], [0], disc=str(c["disc"]))
    return c, pb, XP0


def test_short_runs_match_scipy_step_for_step(golden_ladders):
    c, pb, XP0 = _c1(golden_ladders)
    for rf in (1.0, 1.5 ** 3, 1.5 ** 7):
        res = opt.minimize(lambda z: (lambda r: (r[0], r[3]))(pb.action_grad(z, rf)), XP0,
                           method="L-BFGS-B", jac=True, options=OPTS)
        x, A, st, nit, nfev = pb.minimize_lbfgs(XP0, rf, OPTS)
        assert (nit, nfev, st) == (res.nit, res.nfev, res.status)
        assert abs(A - res.fun) <= 1e-6 * abs(res.fun)
        assert np.abs(x - res.x).max() <= 1e-5


def test_maxiter_and_status_codes(golden_ladders):
    c, pb, XP0 = _c1(golden_ladders)
    o = dict(OPTS, maxiter=5)
    res = opt.minimize(lambda z: (lambda r: (r[0], r[3]))(pb.action_grad(z, 1.5 ** 15)), XP0,
                       method="L-BFGS-B", jac=True, options=o)
    x, A, st, nit, nfev = pb.minimize_lbfgs(XP0, 1.5 ** 15, o)
    assert st == res.status == 1 and nit == res.nit == 5 and nfev == res.nfev
    assert abs(A - res.fun) <= 1e-9 * abs(res.fun)
    # already-converged start: gtol satisfied at x0 -> 0 iterations, status 0
    o = dict(OPTS, gtol=1e3)
    x, A, st, nit, nfev = pb.minimize_lbfgs(XP0, 1.0, o)
    assert (st, nit, nfev) == (0, 0, 1)


def test_ladder_matches_reference_ladder(golden_ladders):
    for name in ("g4_c1_trapezoid_N200", "g4_shipped_SH_N161"):
        c, pb, XP0 = _c1(golden_ladders, name)
        r = pb.anneal(XP0, float(c["alpha"]), c["beta"], OPTS)
        ref_A, ref_k = c["A_array"], c["params"][:, 0]
        # early ladder (few iterations per step): within the optimiser's own
        # stopping threshold (ftol is an ABSOLUTE 1e-8 while A << 1)
        assert np.all(np.abs(r["A"][:12] - ref_A[:12]) <= 1e-8), name
        # end of ladder: same basin, loose (SURVEY.md 8(c) tolerances)
        assert abs(r["A"][-1] - ref_A[-1]) <= 1e-3 * ref_A[-1], name
        assert abs(r["minpaths"][-1, -1] - ref_k[-1]) <= 2e-3 * abs(ref_k[-1]), name
        # iteration counts identical while the run is still short
        assert list(r["nit"][:7]) == list(c["nit"][:7]), name
        # A = me + fe at every stored step (va_ode.py:773-775)
        assert np.allclose(r["A"], r["me"] + r["fe"], rtol=1e-12)


def test_bounded_minimiser_against_scipy():
    """The active-set form of the minimiser (what the device runs with `bounds`; vao_lbfgs_bounded) is
    not L-BFGS-B's Cauchy-point machinery, so its iterates differ from SciPy's; it must stay inside the
    box to the last bit, reproduce the unbounded run when no bound is ever touched, and reach an action
    close to SciPy's L-BFGS-B from the same start."""
    import scipy.optimize as opt
    from varanneal_amd import twin
    D, N = 20, 60
    t, Y, _, Lidx = twin.make_twin(D, N)
    X0, P0 = twin.initial_guess(N, D, 0, Y, Lidx)
    XP = np.append(X0.ravel(), P0)
    pb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P0, [0])
    o = {'gtol': 1e-8, 'ftol': 1e-10, 'maxiter': 2000, 'maxfun': 100000}
    free = pb.minimize_lbfgs(XP, 50.0, o)
    wide = pb.minimize_lbfgs(XP, 50.0, o, bounds=[(-1e6, 1e6)] * (N * D) + [(None, None)])
    assert free[1:] == wide[1:] and np.array_equal(free[0], wide[0])
    fg = lambda x: (lambda r: (r[0], r[3]))(pb.action_grad(x, 50.0))
    for bnds, tol in (([(-15, 15)] * (N * D) + [(6.5, 10.0)], 5e-2), ([(-4.0, 4.0)] * (N * D) + [(7.0, 7.5)], 1e-4),
                      ([(-1.0, 6.0)] * (N * D) + [(None, 8.0)], 1e-4)):
        x, A, st, nit, nfev = pb.minimize_lbfgs(XP, 50.0, o, bounds=bnds)
        rs = opt.minimize(fg, XP, method='L-BFGS-B', jac=True, bounds=bnds, options=o)
        lo = np.array([-np.inf if q[0] is None else q[0] for q in bnds]); hi = np.array([np.inf if q[1] is None else q[1] for q in bnds])
        assert st == 0 and np.all(x >= lo) and np.all(x <= hi)
        assert abs(A - rs.fun) <= tol * rs.fun, (A, rs.fun)
