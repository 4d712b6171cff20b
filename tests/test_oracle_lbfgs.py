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


# ---- L-BFGS-B itself (generalised Cauchy point + subspace minimisation): oracle/va_lbfgsb.inc.c ------------------
def _rosen(x):
    f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g


def _same_as_scipy(fun, x0, bounds, o, xtol=1e-12):
    rs = opt.minimize(fun, x0, jac=True, method='L-BFGS-B', bounds=bounds, options=dict(o, maxcor=10))
    x, f, st, nit, nfev = va_oracle.lbfgs_generic(fun, x0, o, bounds=bounds, exact=True)
    assert (nit, nfev, st) == (rs.nit, rs.nfev, rs.status), ((nit, nfev, st), (rs.nit, rs.nfev, rs.status))
    assert np.abs(x - rs.x).max() <= xtol * max(1.0, np.abs(rs.x).max()), np.abs(x - rs.x).max()
    assert abs(f - rs.fun) <= 1e-13 * max(1.0, abs(rs.fun))
    return rs


def test_lbfgsb_follows_scipy_step_for_step_on_bounded_problems():
    """the restated L-BFGS-B against scipy.optimize.minimize(method='L-BFGS-B', bounds=...) -- the reference's
    minimiser at _autodiffmin.py:85-86 -- iteration for iteration: equal (nit, nfev, status), iterates to 1e-12.
    Boxes, one-sided bounds, mixtures with free variables, iteration caps; problems where tens of breakpoints are
    crossed per Cauchy search and where the subspace step has to be projected back."""
    o = {'gtol': 1e-8, 'ftol': 1e-10, 'maxfun': 10000, 'maxiter': 10000}
    rng = np.random.RandomState(0)
    for n in (2, 5, 20, 50):
        x0 = rng.uniform(-2, 2, n)
        _same_as_scipy(_rosen, x0, [(-1.5, 0.8)] * n, o)
        _same_as_scipy(_rosen, x0, [(None, 0.5) if i % 2 else (0.2, None) for i in range(n)], o)
        _same_as_scipy(_rosen, x0, [(-3, 3) if i % 3 else (None, None) for i in range(n)], dict(o, maxiter=7))
        _same_as_scipy(_rosen, x0, [(0.5, 0.5) if i == 1 else (-1, 2) for i in range(n)], o)        # a variable fixed by l = u
    n = 200
    A = rng.randn(n, n); Q = A.T @ A / n + 0.1 * np.eye(n); b = 3.0 * rng.randn(n)
    quad = lambda x: (0.5 * x @ Q @ x - b @ x, Q @ x - b)
    rs = _same_as_scipy(quad, rng.uniform(-1, 1, n), [(-0.3, 0.3)] * n, o)
    assert np.sum(np.abs(np.abs(rs.x) - 0.3) < 1e-15) > 50                                           # (most variables end on a bound)
    _same_as_scipy(quad, rng.uniform(-1, 1, n), [(-0.3, 0.3) if i % 2 else (0.0, None) for i in range(n)], o)
    # an unbounded problem through the same code (nbd = 0 everywhere): short runs equal SciPy's
    _same_as_scipy(_rosen, rng.uniform(-2, 2, 5), [(None, None)] * 5, dict(o, maxiter=20), xtol=1e-10)


def test_lbfgsb_on_the_action_with_bounds():
    """the Lorenz-96 action with box bounds that bind (the boxes of test_bounded_minimiser_against_scipy): the
    restated L-BFGS-B takes SciPy's iterations"""
    from varanneal_amd import twin
    D, N = 20, 60
    t, Y, _, Lidx = twin.make_twin(D, N)
    X0, P0 = twin.initial_guess(N, D, 0, Y, Lidx)
    XP = np.append(X0.ravel(), P0)
    pb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P0, [0])
    fg = lambda x: (lambda r: (r[0], r[3]))(pb.action_grad(x, 50.0))
    for bnds in ([(-15, 15)] * (N * D) + [(6.5, 10.0)], [(-4.0, 4.0)] * (N * D) + [(7.0, 7.5)],
                 [(-1.0, 6.0)] * (N * D) + [(None, 8.0)]):
        for o in ({'gtol': 1e-8, 'ftol': 1e-10, 'maxiter': 40, 'maxfun': 100000}, {'gtol': 1e-8, 'ftol': 1e-10, 'maxiter': 2000, 'maxfun': 100000}):
            rs = opt.minimize(fg, XP, method='L-BFGS-B', jac=True, bounds=bnds, options=o)
            x, A, st, nit, nfev = pb.minimize_lbfgs(XP, 50.0, o, bounds=bnds, exact=True)
            if o['maxiter'] == 40:
                assert (nit, nfev, st) == (rs.nit, rs.nfev, rs.status)
                assert np.abs(x - rs.x).max() <= 1e-6           # (40 iterations on an ill-conditioned action: rounding grows to ~2e-8)
            else:           # (hundreds of iterations: rounding-level differences grow; same minimum)
                # ftol is ABSOLUTE (1e-10 * max(|f|, 1)) at A ~ 5e-5: both stop on a flat floor, either may be lower
                assert st == rs.status == 0 and abs(A - rs.fun) <= 0.25 * rs.fun
