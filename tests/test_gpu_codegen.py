"""GPU: generated right-hand sides through the C-ABI (va_rhs_load_module + flat tile kernel).
Single evaluations of the tutorial's NaKL neuron model against what the reference produced
(tests/golden/nakl.npz, oracle/gen_golden.py:nakl_cases); a registry-less Lorenz-96 variant
against complex-step derivatives of the NumPy restatement; the bounded NaKL ladder through
the Annealer (reference flow: VarAnneal_tutorial.ipynb NaKL section, va_ode.py:582-605)."""
import numpy as np
import pytest

import va_oracle
from _util import load_npz_cases
from models.nakl import l96_damped, nakl
from varanneal_amd import _capi, codegen, va_ode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nakl_rhs():
    m = codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1)
    return _capi.load_rhs_module(m["so"])


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("nakl.npz")


@pytest.mark.parametrize("name", ["g5_nakl_SimpsonHermite_rf1e+00", "g5_nakl_SimpsonHermite_rf5e+01",
                                  "g5_nakl_trapezoid_rf1e+00", "g5_nakl_trapezoid_rf2e+03"])
def test_nakl_single_eval_matches_reference(nakl_rhs, gold, name):
    c = gold[name]
    D, N = int(c["D"]), int(c["N_model"])
    RF0 = np.resize(c["RF0"], (N - 1, D))
    XP = c["XP"]
    P = XP[N * D:]
    B = 3                                                     # seeds 0 and 2 carry the golden point
    rng = np.random.RandomState(5)
    XPb = np.stack([XP, XP + 0.01 * rng.randn(XP.size), XP])
    pr = _capi.Problem(B, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]), RF0, np.tile(P, (B, 1)),
                       list(range(18)), disc=str(c["disc"]), rhs=nakl_rhs, t_model=c["t"], stim=c["stim"])
    A, me, fe, g = pr.action_grad(XPb, float(c["rf_scale"]))
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * c["A"]
        assert abs(me[b] - c["me"]) <= 1e-12 * c["A"] and abs(fe[b] - c["fe"]) <= 1e-12 * c["A"]
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    assert np.array_equal(g[0], g[2]) and A[1] != A[0]
    pr.close()


@pytest.mark.parametrize("disc", ["trapezoid", "euler", "forwardmap"])
def test_time_dependent_rhs_on_device(disc):
    D, NP, N = 12, 2, 64
    m = codegen.module_for(l96_damped, D, NP)
    rid = _capi.load_rhs_module(m["so"])
    rng = np.random.RandomState(4)
    t = 0.025 * np.arange(N)
    Lidx = [0, 2, 5, 7, 10]
    Y = rng.randn(N, 5)
    P = np.array([8.0, 1.1])
    XP = np.append(3.0 * rng.randn(N * D), P)
    fun = lambda z: va_oracle.numpy_action_generic(l96_damped, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [0, 1], P,
                                                   disc, t_model=t)
    g0 = va_oracle.complex_step_grad(fun, XP)
    pr = _capi.Problem(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None, :], [0, 1], disc=disc, rhs=rid, t_model=t)
    A, me, fe, g = pr.action_grad(XP[None, :], 1.0)
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    pr.close()


def test_nakl_device_minimiser_descends(nakl_rhs, gold):
    """unbounded on-device L-BFGS on a generated RHS: every seed's action decreases and the
    stored minimiser reproduces it through the evaluator."""
    c = gold["g5_nakl_ladder_SH_N101"]
    N, D = int(c["N"]), 4
    RF0 = np.resize(c["RF0"], (N - 1, D))
    B = 4
    rng = np.random.RandomState(1)
    X0 = np.tile(c["X0"].ravel(), (B, 1)) + 1e-3 * rng.randn(B, N * D) * np.arange(B)[:, None]
    X0.reshape(B, N, D)[:, :, 0] = c["Y"][:, 0]
    XP0 = np.hstack([X0, np.tile(c["P0"], (B, 1))])
    pr = _capi.Problem(B, D, N, c["Y"], [0], float(c["t"][1] - c["t"][0]), 1.0, RF0, np.tile(c["P0"], (B, 1)),
                       list(range(18)), disc="SimpsonHermite", rhs=nakl_rhs, t_model=c["t"], stim=c["stim"])
    rf = 1.5 ** 20
    A0 = pr.action_grad(XP0, rf)[0]
    res = pr.minimize_lbfgs(XP0, rf, {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 400, 'maxiter': 400})
    A1 = pr.action_grad(res["x"], rf)[0]
    assert np.all(res["A"] < A0) and np.all(np.abs(A1 - res["A"]) <= 1e-12 * A1)
    pr.close()


def test_nakl_bounded_ladder_matches_reference(gold):
    c = gold["g5_nakl_ladder_SH_N101"]
    a = va_ode.Annealer()
    a.set_model(nakl, 4)
    a.set_data(c["Y"], stim=c["stim"], t=c["t"])
    a.anneal(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), c["beta"], 1.0, list(c["RF0"]), [0],
             list(range(18)), dt_model=None, init_to_data=True, disc="SimpsonHermite", method='L-BFGS-B',
             bounds=[tuple(b) for b in c["bounds"]],
             opt_args={'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}, adolcID=0,
             verbose=False)
    # same SciPy driver around an evaluator that agrees to ~1e-13: the first ladder steps follow
    # the reference iterate for iterate, later ones drift with the accumulated rounding
    assert list(a.nit_array[:4]) == list(c["nit"][:4])
    assert np.all(np.abs(a.A_array[:7] - c["A_array"][:7]) <= 1e-5 * c["A_array"][:7])
    assert np.all(np.abs(a.A_array - c["A_array"]) <= 2e-2 * c["A_array"])
    lo, hi = c["bounds"][4:, 0], c["bounds"][4:, 1]
    assert np.all(a.P >= lo - 1e-12) and np.all(a.P <= hi + 1e-12)
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    a.close()


def test_nakl_bounded_ladder_on_the_device_batched(gold):
    """Box bounds on the device (bounded_minimiser='device'): 16 seeds of the tutorial's bounded NaKL ladder anneal in
    ONE call, nothing leaves HBM.  The device runs L-BFGS-B itself (generalised Cauchy point + subspace minimisation,
    csrc/va_lbfgsb.hip), so the unperturbed seed follows the reference's own SciPy ladder (tests/golden/nakl.npz):
    its iteration counts on the first rungs and its action on every rung; every stored iterate is inside the box."""
    c = gold["g5_nakl_ladder_SH_N101"]
    N, D, B = int(c["N"]), 4, 16
    rng = np.random.RandomState(3)
    X0 = np.tile(c["X0"], (B, 1, 1)); P0 = np.tile(c["P0"], (B, 1))
    X0[1:] += 1e-2 * rng.randn(B - 1, N, D); P0[1:] *= 1.0 + 1e-2 * rng.randn(B - 1, P0.shape[1])
    bnds = [tuple(b) for b in c["bounds"]]
    lo = np.array([b[0] for b in bnds]); hi = np.array([b[1] for b in bnds])
    P0 = np.clip(P0, lo[4:], hi[4:])
    a = va_ode.Annealer()
    a.set_model(nakl, 4)
    a.set_data(c["Y"], stim=c["stim"], t=c["t"])
    a.anneal(X0, P0, float(c["alpha"]), c["beta"], 1.0, list(c["RF0"]), [0], list(range(18)), dt_model=None,
             init_to_data=True, disc="SimpsonHermite", method='L-BFGS-B', bounds=bnds,
             opt_args={'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}, adolcID=0, verbose=False)
    assert a._device_bounds and a.A_array.shape == (B, len(c["beta"]))
    X = a.minpaths[:, :, :N * D].reshape(B, -1, N, D); P = a.minpaths[:, :, N * D:]
    assert np.all(X >= lo[:4]) and np.all(X <= hi[:4]) and np.all(P >= lo[4:]) and np.all(P <= hi[4:])
    assert np.all(a.exitflags == 0)
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    rel = np.abs(a.A_array[0] - c["A_array"]) / c["A_array"]
    print("bounded NaKL ladder on the device, seed 0 vs the reference's SciPy ladder: rel. deviation per rung",
          np.array2string(rel, precision=2), " iterations", a.nit_array[0], "reference", c["nit"])
    # measured: iterations [38 1 1 1 62 5 5 58] against the reference's [38 1 1 1 62 5 5 86]; deviations 1e-10 ... 9e-8 on
    # the first seven rungs, 1.7e-4 on the last (ftol = 1e-8 is absolute at A ~ 1e-3: both stop on a flat floor)
    assert list(a.nit_array[0][:7]) == list(c["nit"][:7])
    assert np.all(rel[:7] <= 1e-6) and rel[7] <= 1e-3
    a.close()


def test_bounded_device_minimiser_is_lbfgsb_step_for_step():
    """The device's bounded minimiser against L-BFGS-B itself -- the oracle's restatement (vao_lbfgsb), which
    tests/test_oracle_lbfgs.py pins to scipy.optimize.minimize step for step, and SciPy directly: identical
    (nit, nfev, status) and the same iterate after 25 iterations, on boxes that bind, one-sided bounds and a free parameter."""
    import scipy.optimize as opt
    import va_oracle
    from varanneal_amd import twin
    D, N, B = 20, 60, 3
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    for o in ({'gtol': 1e-8, 'ftol': 1e-10, 'maxiter': 25, 'maxfun': 100000}, {'gtol': 1e-8, 'ftol': 1e-10, 'maxiter': 3, 'maxfun': 100000}):
        for bnds in ([(-4.0, 4.0)] * (N * D) + [(7.0, 7.5)], [(-1.0, 6.0)] * (N * D) + [(None, 8.0)],
                     [(-15, 15)] * (N * D) + [(6.5, 10.0)], [(None, 5.0) if i % 2 else (-5.0, None) for i in range(N * D)] + [(None, None)]):
            lo = np.array([-np.inf if q[0] is None else q[0] for q in bnds]); hi = np.array([np.inf if q[1] is None else q[1] for q in bnds])
            with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", bounds=bnds) as pb:
                r = pb.minimize_lbfgs(XP, 50.0, o)
            assert np.all(r["x"] >= lo) and np.all(r["x"] <= hi)
            for b in range(B):
                opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0])
                x, A, st, nit, nfev = opb.minimize_lbfgs(XP[b], 50.0, o, bounds=bnds, exact=True)
                assert (r["nit"][b], r["nfev"][b], r["status"][b]) == (nit, nfev, st), (b, o["maxiter"], r["nit"][b], nit, r["nfev"][b], nfev)
                assert abs(r["A"][b] - A) <= 1e-9 * abs(A) and np.abs(r["x"][b] - x).max() <= 1e-6
            fg = lambda z: (lambda q: (q[0], q[3]))(opb.action_grad(z, 50.0))
            rs = opt.minimize(fg, XP[B - 1], method='L-BFGS-B', jac=True, bounds=bnds, options=o)
            assert (r["nit"][B - 1], r["nfev"][B - 1], r["status"][B - 1]) == (rs.nit, rs.nfev, rs.status)
            assert np.abs(r["x"][B - 1] - rs.x).max() <= 1e-6


def test_bounded_device_minimiser_random_boxes():
    """a seeded sweep of box shapes for the device's L-BFGS-B: random finite / one-sided / absent bounds per variable,
    variables fixed by l = u, starting points outside the box (projected first), Simpson-Hermite and trapezoid, several
    widths -- (nit, nfev, status) and the end point of the oracle's L-BFGS-B, which is SciPy's step for step"""
    import va_oracle
    from varanneal_amd import twin
    rng = np.random.RandomState(77)
    for case in range(8):
        D = int(rng.choice([6, 10, 20]))
        disc = "SimpsonHermite" if case % 3 == 2 else "trapezoid"
        N = int(rng.randint(20, 45)) | (1 if disc == "SimpsonHermite" else 0)
        B = 2
        t, Y, _, Lidx = twin.make_twin(D, N)
        XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
        for b in range(B):
            X0, P0 = twin.initial_guess(N, D, b + case, Y, Lidx)
            XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
        bnds = []
        for i in range(N * D + 1):
            kind = rng.randint(0, 6)
            c = XP[0, i]
            if kind == 0: bnds.append((None, None))
            elif kind == 1: bnds.append((c - rng.rand() * 2.0, None))
            elif kind == 2: bnds.append((None, c + rng.rand() * 2.0))
            elif kind == 3: bnds.append((c + 0.1, c + 0.1 + rng.rand()))           # the start lies outside
            elif kind == 4 and i % 7 == 0: bnds.append((c, c))                     # fixed
            else: bnds.append((c - rng.rand() * 3.0, c + rng.rand() * 3.0))
        o = {'gtol': 1e-8, 'ftol': 1e-10, 'maxiter': int(rng.choice([4, 12, 30])), 'maxfun': 100000}
        with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc=disc, bounds=bnds) as pb:
            r = pb.minimize_lbfgs(XP, 20.0, o)
        lo = np.array([-np.inf if q[0] is None else q[0] for q in bnds]); hi = np.array([np.inf if q[1] is None else q[1] for q in bnds])
        assert np.all(r["x"] >= lo) and np.all(r["x"] <= hi), case
        for b in range(B):
            opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc=disc)
            x, A, st, nit, nfev = opb.minimize_lbfgs(XP[b], 20.0, o, bounds=bnds, exact=True)
            assert (r["nit"][b], r["nfev"][b], r["status"][b]) == (nit, nfev, st), (case, b, D, N, disc, o["maxiter"], r["nit"][b], nit, r["nfev"][b], nfev)
            # (same iteration and evaluation counts; rounding differences between the two arithmetic orders grow along an
            #  ill-conditioned run: 1.6e-8 on A after 4-30 iterations here)
            assert abs(r["A"][b] - A) <= 1e-6 * abs(A) and np.abs(r["x"][b] - x).max() <= 1e-5, (case, b, abs(r["A"][b] - A) / abs(A))


# ---- generated models on the wave-private column-run kernel (codegen.column_form, va_eval_plan) ---------------
def _l96_user(t, x, p):
    """Lorenz-96 as a user would write it (not the registry's callable): traced, recognised as a stencil"""
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - x + p[0]


@pytest.mark.parametrize("name", ["g5_nakl_SimpsonHermite_rf1e+00", "g5_nakl_SimpsonHermite_rf5e+01",
                                  "g5_nakl_trapezoid_rf1e+00", "g5_nakl_trapezoid_rf2e+03"])
def test_nakl_on_the_column_kernel_matches_reference(gold, name):
    """the reference's own NaKL numbers again, now with the model's dense column form on k_eval4"""
    c = gold[name]
    D, N = int(c["D"]), int(c["N_model"])
    RF0 = np.resize(c["RF0"], (N - 1, D))
    XP = c["XP"]
    P = XP[N * D:]
    B = 3
    m = codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1,
                           col_variant=lambda ne, gh: _capi.eval_plan(B, D, N, str(c["disc"]), ne, gh, rf_array=True))
    assert m["col_variant"] is not None
    rng = np.random.RandomState(5)
    XPb = np.stack([XP, XP + 0.01 * rng.randn(XP.size), XP])
    pr = _capi.Problem(B, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]), RF0, np.tile(P, (B, 1)),
                       list(range(18)), disc=str(c["disc"]), rhs=_capi.load_rhs_module(m["so"]), t_model=c["t"],
                       stim=c["stim"])
    assert pr.info()["eval_kernel"] == 4 and pr.info()["run_rows"] == m["col_variant"][2]
    A, me, fe, g = pr.action_grad(XPb, float(c["rf_scale"]))
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * c["A"]
        assert abs(me[b] - c["me"]) <= 1e-12 * c["A"] and abs(fe[b] - c["fe"]) <= 1e-12 * c["A"]
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    assert np.array_equal(g[0], g[2]) and A[1] != A[0]
    pr.close()


def test_column_module_on_another_geometry_runs_its_flat_kernel(gold):
    """a module carries ONE instantiation of k_eval4; a problem that calls for another runs the flat kernel"""
    c = gold["g5_nakl_trapezoid_rf1e+00"]
    D, N = int(c["D"]), int(c["N_model"])
    m = codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1, col_variant=lambda ne, gh: (4, 2, 4, 0))   # Simpson-Hermite
    XP = c["XP"]; P = XP[N * D:]
    pr = _capi.Problem(1, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]), np.resize(c["RF0"], (N - 1, D)),
                       P[None, :], list(range(18)), disc="trapezoid", rhs=_capi.load_rhs_module(m["so"]),
                       t_model=c["t"], stim=c["stim"])
    assert pr.info()["eval_kernel"] == 1
    A, me, fe, g = pr.action_grad(XP[None, :], float(c["rf_scale"]))
    assert abs(A[0] - c["A"]) <= 1e-12 * c["A"] and np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    pr.close()


def test_traced_l96_runs_at_the_builtin_speed_and_agrees():
    """a traced Lorenz-96 (not recognised by the registry) at BASELINE config 3's shape:
    within 1e-13 of the built-in's values and within 10 % of its evaluation time (same kernel, generated RHS)."""
    from varanneal_amd import twin
    D, N, B = 20, 1000, 64
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1))
    for s in range(B):
        x, p = twin.initial_guess(N, D, s, Y, Lidx)
        XP[s, :N * D] = x.ravel(); XP[s, N * D:] = p
    P = XP[:, N * D:].copy()
    m = codegen.module_for(_l96_user, D, 1, col_variant=lambda ne, gh: _capi.eval_plan(B, D, N, "trapezoid", ne, gh))
    assert m["col"]["uniform"] and m["col_variant"] == (4, 1, 7, 1)
    out, us = {}, {}
    for rhs in ("lorenz96", _capi.load_rhs_module(m["so"])):
        pr = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 4e-6, P, [0], disc="trapezoid", rhs=rhs)
        assert pr.info()["eval_kernel"] == 4 and pr.info()["run_rows"] == 7
        out[rhs] = pr.action_grad(XP, 1.5 ** 20)
        pr.eval_timed(1.5 ** 20, 200)                                   # warm-up
        us[rhs] = min(pr.eval_timed(1.5 ** 20, 2000) for _ in range(5)) / 2000 * 1e3
        pr.close()
    (Ab, meb, feb, gb), (A, me, fe, g) = out["lorenz96"], out[[k for k in out if k != "lorenz96"][0]]
    assert np.all(np.abs(A - Ab) <= 1e-13 * np.abs(Ab)) and np.all(np.abs(fe - feb) <= 1e-13 * np.abs(Ab))
    assert np.array_equal(me, meb)
    assert np.abs(g - gb).max() <= 1e-13 * np.abs(gb).max()
    ub, uu = us["lorenz96"], [v for k, v in us.items() if k != "lorenz96"][0]
    # (recorded, not asserted: box-to-box spread is several per cent; tools/sweep.sh times both under profiles/)
    print("C3 evaluation: built-in %.2f us, traced + generated %.2f us" % (ub, uu))


def test_annealer_puts_a_traced_stencil_on_the_column_kernel():
    """through the drop-in: set_model(callable) -> trace -> column form -> the module instantiation the
    problem's geometry calls for; one rung against the built-in's"""
    from varanneal_amd import twin
    D, N, B = 20, 300, 8
    t, Y, _, Lidx = twin.make_twin(D, N)
    X0 = np.empty((B, N, D)); P0 = np.empty((B, 1))
    for s in range(B):
        X0[s], P0[s] = twin.initial_guess(N, D, s, Y, Lidx)
    res = {}
    for f in (twin.l96, _l96_user):
        a = va_ode.Annealer()
        a.set_model(f, D)
        a.set_data(Y, t=t)
        a.anneal(X0.copy(), P0.copy(), 1.5, [0], 4.0, 4e-6, list(Lidx), [0], disc="trapezoid",
                 opt_args={'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': 25, 'maxfun': 1000}, verbose=False)
        assert a._pb.info()["eval_kernel"] == 4
        res[f] = (a.A_array.copy(), a.nit_array.copy())
        a.close()
    assert np.array_equal(res[twin.l96][1], res[_l96_user][1])
    assert np.allclose(res[twin.l96][0], res[_l96_user][0], rtol=1e-9)


# ---- wide generated stencils on the workgroup column-run kernel (codegen.ghost_form) ------------------------------
def _wide_stencil(t, x, p):
    """not Lorenz-96: five-point stencil, cubic damping, two parameters; its adjoint needs 4 ghost columns"""
    return (np.roll(x, 1, 1) * (np.roll(x, -2, 1) - np.roll(x, 2, 1)) - p[1] * x ** 3
            + p[0] * np.roll(x, -1, 1))


def test_traced_l96_at_c4_width_runs_the_builtin_kernels():
    """D = 200 (BASELINE config 4's width): the traced Lorenz-96 gets the kernel the built-in runs -- the streaming
    kernel k_eval5 through its column form, one-step rules and Simpson-Hermite -- with values within 1e-13 of the
    built-in's"""
    from varanneal_amd import twin
    D, B = 200, 16
    Lidx = list(range(0, D, 5))
    for disc, N, ek in (("trapezoid", 2000, 5), ("SimpsonHermite", 1001, 5)):
        t, Y, _, _ = twin.make_twin(D, N, Lidx=Lidx)
        rng = np.random.RandomState(0)
        XP = np.concatenate([8.0 * rng.rand(B, N * D) - 4.0, 8.17 + 0.1 * rng.randn(B, 1)], axis=1)
        P = XP[:, N * D:].copy()
        m = codegen.module_for(_l96_user, D, 1, col_variant=lambda ne, gh, reach: _capi.eval_plan(B, D, N, disc, ne, gh, reach=reach, Lidx=Lidx))
        assert m["col_variant"][0] == ek, m["col_variant"]
        assert (m["ghost"] is not None and m["ghost"]["GHOST"] == 2) if ek == 3 else (m["col"] is not None and m["col"]["reach"] == (2, 1, 1, 2))
        out, us = {}, {}
        for key, rhs in (("builtin", "lorenz96"), ("traced", _capi.load_rhs_module(m["so"]))):
            pr = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 4e-6, P, [0], disc=disc, rhs=rhs)
            assert pr.info()["eval_kernel"] == ek
            out[key] = pr.action_grad(XP, 1.5 ** 20)
            pr.eval_timed(1.5 ** 20, 20)
            us[key] = min(pr.eval_timed(1.5 ** 20, 100) for _ in range(3)) / 100 * 1e3
            pr.close()
        (Ab, meb, feb, gb), (A, me, fe, g) = out["builtin"], out["traced"]
        assert np.all(np.abs(A - Ab) <= 1e-13 * np.abs(Ab)) and np.all(np.abs(me - meb) <= 1e-13 * np.abs(Ab))
        assert np.abs(g - gb).max() <= 1e-13 * np.abs(gb).max()
        print("D=200 N=%d B=%d %s evaluation: built-in %.1f us, traced + generated %.1f us" % (N, B, disc, us["builtin"], us["traced"]))


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite"])
def test_wide_stencil_with_four_ghost_columns_on_device(disc):
    D, NP, N, B = 70, 2, 201, 3
    rng = np.random.RandomState(70)
    Lidx = list(range(0, D, 7))
    Y = rng.randn(N, len(Lidx)); P = np.array([8.0, 0.05])
    XP = np.append(2.0 * rng.randn(N * D), P)
    m = codegen.module_for(_wide_stencil, D, NP, col_variant=lambda ne, gh: _capi.eval_plan(B, D, N, disc, ne, gh))
    assert m["ghost"]["GHOST"] == 4 and m["col_variant"][0] == 3
    fun = lambda z: va_oracle.numpy_action_generic(_wide_stencil, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [0, 1], P, disc)
    pr = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 0.3, np.tile(P, (B, 1)), [0, 1], disc=disc,
                       rhs=_capi.load_rhs_module(m["so"]))
    assert pr.info()["eval_kernel"] == 3
    XPb = np.stack([XP, XP * 1.01, XP])
    A, me, fe, g = pr.action_grad(XPb, 1.0)
    # the flat kernel of the same module on the same point (independent mapping, same generated derivatives)
    pf = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 0.3, np.tile(P, (B, 1)), [0, 1], disc=disc,
                       rhs=_capi.load_rhs_module(m["so"]), eval_kernel=1)
    assert pf.info()["eval_kernel"] == 1
    Af, mef, fef, gf = pf.action_grad(XPb, 1.0)
    A0 = fun(XP)[0]
    assert abs(A[0] - A0) <= 1e-12 * A0 and abs(A[2] - A0) <= 1e-12 * A0
    assert np.abs(g - gf).max() <= 1e-12 * np.abs(gf).max()
    # a handful of gradient entries against complex-step derivatives of the NumPy restatement
    idx = np.r_[rng.choice(N * D, 12, replace=False), N * D, N * D + 1]
    z = XP.astype(complex)
    for i in idx:
        z[i] += 1e-30j
        gi = fun(z)[0].imag / 1e-30
        z[i] = XP[i]
        assert abs(g[0, i] - gi) <= 1e-10 * np.abs(g[0]).max()
    pr.close(); pf.close()


def _stencil_5p(t, x, p):
    """five parameters: more partial-sum columns than one 8-value reduction row of the workgroup kernel holds"""
    return (p[0] * np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[1] * x + p[2]
            + p[3] * np.roll(x, -1, 1) - p[4] * x ** 2)


def test_ghosted_stencil_with_five_parameters_on_device():
    D, NP, N, B = 100, 5, 120, 2
    rng = np.random.RandomState(100)
    Lidx = list(range(0, D, 9))
    Y = rng.randn(N, len(Lidx)); P = np.array([1.0, 1.0, 8.0, 0.1, 0.02])
    XP = np.append(2.0 * rng.randn(N * D), P)
    m = codegen.module_for(_stencil_5p, D, NP, col_variant=lambda ne, gh: _capi.eval_plan(B, D, N, "trapezoid", ne, gh))
    assert m["ghost"] is not None and m["col_variant"][0] == 3
    rid = _capi.load_rhs_module(m["so"])
    out = {}
    for ek in (0, 1):
        pr = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 0.3, np.tile(P, (B, 1)), list(range(NP)), disc="trapezoid",
                           rhs=rid, eval_kernel=ek)
        assert pr.info()["eval_kernel"] == (3 if ek == 0 else 1)
        out[ek] = pr.action_grad(np.stack([XP, XP * 0.99]), 2.0)
        pr.close()
    fun = lambda z: va_oracle.numpy_action_generic(_stencil_5p, z, D, N, Y, Lidx, 0.025, 4.0, 0.3 * 2.0, NP, list(range(NP)),
                                                   P, "trapezoid")
    A0 = fun(XP)[0]
    for ek in (0, 1):
        assert abs(out[ek][0][0] - A0) <= 1e-12 * A0
    assert np.abs(out[0][3] - out[1][3]).max() <= 1e-12 * np.abs(out[1][3]).max()
    z = XP.astype(complex)
    for i in range(N * D, N * D + NP):                      # the five parameter derivatives against complex step
        z[i] += 1e-30j
        gi = fun(z)[0].imag / 1e-30
        z[i] = XP[i]
        assert abs(out[0][3][0, i] - gi) <= 1e-10 * np.abs(out[0][3][0]).max(), (i, out[0][3][0, i], gi)


# ---- block-periodic models (a ring of identical units): periodic column form on the column-run kernels ------------
def _ring_of_units(t, x, p):
    """five identical 4-state units on a ring (D = 20): NOT a stencil (see tests/test_codegen.py)"""
    D = x.shape[-1]
    U = D // 4
    out = []
    for u in range(U):
        v, a, b, c = x[..., 4 * u], x[..., 4 * u + 1], x[..., 4 * u + 2], x[..., 4 * u + 3]
        vl, vr = x[..., 4 * ((u - 1) % U)], x[..., 4 * ((u + 1) % U)]
        out.append(p[0] * (vl + vr - 2.0 * v) - v * v * v + a * v - b + c * c)
        out.append(p[1] * (v - a))
        out.append(p[2] * (v * v - b))
        out.append(-c + 0.5 * v * a)
    return np.stack(out, axis=-1)


def test_ring_of_units_runs_the_column_kernel_at_c3_shape():
    """a D = 20 model that is neither a stencil nor small: its periodic column form (4 classes) runs k_eval4; values
    and gradient against the oracle's generic NumPy action (complex-step derivative) to 1e-12 / 1e-10, and against the
    same module's flat kernel; time per evaluation of 64 seeds recorded next to the built-in Lorenz-96's"""
    from varanneal_amd import twin
    D, N, B, NP = 20, 1000, 64, 3
    t, Y, _, Lidx = twin.make_twin(D, N)
    rng = np.random.RandomState(8)
    P = np.array([0.7, 0.9, 1.3])
    XP = np.concatenate([0.8 * rng.randn(B, N * D), np.tile(P, (B, 1))], axis=1)
    m = codegen.module_for(_ring_of_units, D, NP, col_variant=lambda ne, gh: _capi.eval_plan(B, D, N, "trapezoid", ne, gh))
    assert m["col"] is not None and m["col"]["period"] == 4 and m["col_variant"][0] == 4
    rid = _capi.load_rhs_module(m["so"])
    us = {}
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 0.3, np.tile(P, (B, 1)), [0, 1, 2], disc="trapezoid", rhs=rid) as pr:
        assert pr.info()["eval_kernel"] == 4
        A, me, fe, g = pr.action_grad(XP, 2.0)
        pr.eval_timed(2.0, 100)
        us["ring, column form"] = min(pr.eval_timed(2.0, 1000) for _ in range(3))
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 0.3, np.tile(P, (B, 1)), [0, 1, 2], disc="trapezoid", rhs=rid, eval_kernel=1) as pf:
        assert pf.info()["eval_kernel"] == 1
        Af, mef, fef, gf = pf.action_grad(XP, 2.0)
        pf.eval_timed(2.0, 20)
        us["ring, flat form"] = min(pf.eval_timed(2.0, 200) for _ in range(3)) * 5
    P1 = XP[:, -1:].copy() * 0 + 8.0
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 0.3, P1, [0], disc="trapezoid") as pl:
        pl.action_grad(np.concatenate([XP[:, :N * D], P1], axis=1), 2.0)
        pl.eval_timed(2.0, 100)
        us["built-in Lorenz-96"] = min(pl.eval_timed(2.0, 1000) for _ in range(3))
    print("column vs flat form: A rel", (np.abs(A - Af) / np.abs(Af)).max(), "grad rel", np.abs(g - gf).max() / np.abs(gf).max())
    assert np.all(np.abs(A - Af) <= 1e-12 * np.abs(Af)) and np.abs(g - gf).max() <= 1e-11 * np.abs(gf).max()
    fun = lambda z: va_oracle.numpy_action_generic(_ring_of_units, z, D, N, Y, Lidx, twin.DT, 4.0, 0.3 * 2.0, NP, [0, 1, 2], P, "trapezoid")
    for b in (0, B - 1):
        A0 = fun(XP[b])[0]
        assert abs(A[b] - A0) <= 1e-12 * abs(A0)
    g0 = va_oracle.complex_step_grad(fun, XP[3])
    assert np.abs(g[3] - g0).max() <= 1e-10 * np.abs(g0).max()
    print("D = 20, N = 1000, 64 seeds, us per complete evaluation:", {k: round(v, 2) for k, v in us.items()})


# ---- dense constant linear part on the matrix cores (codegen.linear_split, va_eval_flat.h lin_gemm) -----------------
def _coupled(D, seed=0):
    """dense linear coupling (a fixed D x D matrix) + local cubic damping + a drive: neither a stencil nor small"""
    C = np.random.RandomState(seed).randn(D, D) / np.sqrt(D)

    def coupled(t, x, p):
        return x @ C.T - p[1] * x ** 3 + p[0]
    return coupled, C


@pytest.mark.parametrize("D,N", [(20, 61), (48, 40), (50, 33), (200, 12)])
def test_dense_linear_part_matches_complex_step(D, N):
    """f = C x + rest(x, p): the generator splits C off, the flat kernel forms X C^T and S C with
    v_mfma_f64_16x16x4_f64 (BASELINE north_star: MFMA for a dense D x D linear map) and the element-wise code keeps the
    rest; A to 1e-12 and the gradient to 1e-10 of the oracle's generic NumPy action with complex-step derivatives,
    all four discretisations, widths that are / are not multiples of 16"""
    f, C = _coupled(D, D)
    m = codegen.module_for(f, D, 2)
    assert m["lin"] is not None and np.array_equal(m["lin"], C) and "LINEAR = true" in m["text"]
    rid = _capi.load_rhs_module(m["so"])
    rng = np.random.RandomState(N)
    B = 3
    Lidx = sorted(rng.choice(D, max(2, D // 4), replace=False).tolist())
    Y = rng.randn(N, len(Lidx))
    P = np.array([1.5, 0.3])
    XP = np.concatenate([rng.randn(B, N * D), np.tile(P, (B, 1))], axis=1)
    for disc in (("trapezoid", "SimpsonHermite", "euler", "forwardmap") if D < 200 else ("trapezoid",)):
        Nd = N + (disc == "SimpsonHermite" and N % 2 == 0)
        Yd = Y if Nd == N else np.vstack([Y, Y[-1:]])
        XPd = XP if Nd == N else np.concatenate([rng.randn(B, Nd * D), np.tile(P, (B, 1))], axis=1)
        with _capi.Problem(B, D, Nd, Yd, Lidx, 0.02, 2.0, 0.7, np.tile(P, (B, 1)), [0, 1], disc=disc, rhs=rid) as pr:
            assert pr.info()["eval_kernel"] == 1
            A, me, fe, g = pr.action_grad(XPd, 3.0)
        fun = lambda z: va_oracle.numpy_action_generic(f, z, D, Nd, Yd, Lidx, 0.02, 2.0, 0.7 * 3.0, 2, [0, 1], P, disc)
        for b in range(B):
            A0, me0, fe0 = fun(XPd[b])
            assert abs(A[b] - A0) <= 1e-12 * abs(A0) and abs(me[b] - me0) <= 1e-12 * abs(A0), (D, disc, b)
        g0 = va_oracle.complex_step_grad(fun, XPd[1])
        assert np.abs(g[1] - g0).max() <= 1e-10 * np.abs(g0).max(), (D, disc)


def test_dense_linear_part_at_c3_shape_timed():
    """D = 20, N = 1000, 64 seeds: the split module against the same model with the linear part left in the
    element-wise code (a 20-term switch per column) -- same values, time per evaluation recorded"""
    from varanneal_amd import twin
    D, N, B = 20, 1000, 64
    f, C = _coupled(D, 1)
    t, Y, _, Lidx = twin.make_twin(D, N)
    rng = np.random.RandomState(2)
    P = np.array([1.5, 0.3])
    XP = np.concatenate([rng.randn(B, N * D), np.tile(P, (B, 1))], axis=1)
    out, us = {}, {}
    for key, lin in (("matrix cores", True), ("element-wise", False)):
        m = codegen.module_for(f, D, 2, linear=lin)
        assert (m["lin"] is not None) == lin
        with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 0.3, np.tile(P, (B, 1)), [0, 1], disc="trapezoid",
                           rhs=_capi.load_rhs_module(m["so"])) as pr:
            out[key] = pr.action_grad(XP, 2.0)
            pr.eval_timed(2.0, 20)
            us[key] = min(pr.eval_timed(2.0, 200) for _ in range(3)) * 5
    (A, me, fe, g), (Ae, mee, fee, ge) = out["matrix cores"], out["element-wise"]
    assert np.all(np.abs(A - Ae) <= 1e-12 * np.abs(Ae)) and np.abs(g - ge).max() <= 1e-11 * np.abs(ge).max()
    A0 = va_oracle.numpy_action_generic(f, XP[5], D, N, Y, Lidx, twin.DT, 4.0, 0.3 * 2.0, 2, [0, 1], P, "trapezoid")[0]
    assert abs(A[5] - A0) <= 1e-12 * abs(A0)
    print("dense coupling D = 20, N = 1000, 64 seeds, us per complete evaluation:", {k: round(v, 2) for k, v in us.items()})


def test_dense_linear_part_with_stimulus_and_time():
    """the split model of tests/test_codegen.py::test_linear_module_with_stimulus_and_explicit_time on the device:
    X C^T and S C on the matrix cores, the stimulus / time terms in the element-wise rest"""
    D, NP, N, B = 16, 2, 25, 2
    C = np.random.RandomState(3).randn(D, D) / 4.0

    def driven(t, x, ps):
        p, stim = ps
        drive = p[0] * stim * np.cos(0.3 * t)
        return x @ C.T - p[1] * x ** 3 + drive[:, None]
    m = codegen.module_for(driven, D, NP, nstim=1, stim_ndim=1)
    assert m["lin"] is not None
    rng = np.random.RandomState(0)
    t = 0.05 * np.arange(N)
    stim = np.sin(1.3 * t) + 0.1 * rng.randn(N)
    Y = rng.randn(N, 4); Lidx = [1, 4, 9, 15]
    P = np.array([0.8, 0.2])
    XP = np.concatenate([rng.randn(B, N * D), np.tile(P, (B, 1))], axis=1)
    for disc in ("trapezoid", "SimpsonHermite", "euler"):
        with _capi.Problem(B, D, N, Y, Lidx, 0.05, 2.0, 0.5, np.tile(P, (B, 1)), [0, 1], disc=disc, rhs=_capi.load_rhs_module(m["so"]),
                           t_model=t, stim=stim) as pr:
            A, me, fe, g = pr.action_grad(XP, 1.0)
        fun = lambda z: va_oracle.numpy_action_generic(driven, z, D, N, Y, Lidx, 0.05, 2.0, 0.5, NP, [0, 1], P, disc, t_model=t, stim=stim)
        for b in range(B):
            assert abs(A[b] - fun(XP[b])[0]) <= 1e-12 * abs(A[b]), (disc, b)
        g0 = va_oracle.complex_step_grad(fun, XP[1])
        assert np.abs(g[1] - g0).max() <= 1e-10 * np.abs(g0).max(), disc


def _many_parameters(t, x, p):
    """40 parameters: every state its own forcing p[i] and damping p[20 + i] (the reference has no cap on NP,
    varanneal/va_ode.py:564-578; the tuned kernels and the rows of partial sums carry 24)"""
    D = x.shape[1]
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[D:2 * D] * x + p[:D]


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler"])
def test_forty_parameter_model_on_device(disc):
    """single evaluations against complex-step derivatives of the NumPy restatement with the ORIGINAL callable, estimated
    parameters on both sides of the 24th; then a short minimisation that must descend"""
    D, NP, N, B = 20, 40, 201, 3
    m = codegen.module_for(_many_parameters, D, NP)
    rid = _capi.load_rhs_module(m["so"])
    rng = np.random.RandomState(9)
    Lidx = [0, 3, 7, 11, 14, 18]
    Y = rng.randn(N, len(Lidx))
    P = np.append(8.0 + rng.rand(D), 0.8 + 0.4 * rng.rand(D))
    Pidx = [0, 5, 19, 20, 23, 24, 25, 31, 39]
    XP = np.stack([np.append(3.0 * rng.randn(N * D), P[Pidx]) for _ in range(B)])
    rf = 20.0
    with _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 0.01, np.tile(P, (B, 1)), Pidx, disc=disc, rhs=rid) as pr:
        assert pr.info()["eval_kernel"] == 1 and pr.persistent() is None
        A, me, fe, g = pr.action_grad(XP, rf)
        for b in range(B):
            fun = lambda z: va_oracle.numpy_action_generic(_many_parameters, z, D, N, Y, Lidx, 0.025, 4.0, 0.01 * rf, NP, Pidx, P, disc)
            A0 = fun(XP[b])[0]
            assert abs(A[b] - A0) <= 1e-12 * abs(A0)
            if b == 0:
                g0 = va_oracle.complex_step_grad(fun, XP[b])
                assert np.abs(g[b] - g0).max() <= 1e-10 * np.abs(g0).max()
                assert np.abs(g[b, N * D:] - g0[N * D:]).max() <= 1e-10 * np.abs(g0[N * D:]).max()      # (the parameter block on its own scale)
        r = pr.minimize_lbfgs(XP, rf, {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 300, 'maxiter': 200})
        assert np.all(r["A"] < A) and np.all(np.isfinite(r["x"]))
        A1 = pr.action_grad(r["x"], rf)[0]
        assert np.all(np.abs(A1 - r["A"]) <= 1e-12 * A1)


def _relu_ring(t, x, p):
    """a piecewise model: rectified coupling, two leak rates chosen by the state, a saturating neighbour term -- what the
    reference would tape as it executes (_autodiffmin.py:41-44); traced to selects (codegen.SymArray)"""
    drive = np.maximum(np.roll(x, 1, 1), 0.0)
    leak = np.where(x > 0.5, p[1], 0.5 * p[1])
    return -leak * x + p[0] * drive + 0.3 * np.clip(np.roll(x, -1, 1), -1.0, 1.5) - 0.2 * np.minimum(x, np.roll(x, 2, 1))


@pytest.mark.parametrize("D,N,disc", [(10, 200, "trapezoid"), (10, 201, "SimpsonHermite"), (20, 1000, "trapezoid"), (100, 300, "euler")])
def test_piecewise_model_on_device(D, N, disc):
    """value and gradient (the derivative of the taken branch) against complex-step through the original callable, on the
    kernels the plan picks (column kernel for narrow states, streaming / flat beyond) and on the flat kernel"""
    NP = 2
    rng = np.random.RandomState(7)
    Lidx = list(range(0, D, 3))
    Y = rng.randn(N, len(Lidx))
    P = np.array([1.3, 0.8]); Pidx = [0, 1]
    XP = np.append(1.5 * rng.randn(N * D), P)[None, :]
    rf = 30.0
    fun = lambda z: va_oracle.numpy_action_generic(_relu_ring, z, D, N, Y, Lidx, 0.05, 2.0, 0.1 * rf, NP, Pidx, P, disc)
    A0 = fun(XP[0])[0]
    g0 = va_oracle.complex_step_grad(fun, XP[0])
    kernels = set()
    for ek in (0, 1):
        m = codegen.module_for(_relu_ring, D, NP, col_variant=lambda ne, gh, reach: _capi.eval_plan(
            1, D, N, disc, ne, gh, eval_kernel=ek, reach=reach, Lidx=Lidx))
        rid = _capi.load_rhs_module(m["so"])
        with _capi.Problem(1, D, N, Y, Lidx, 0.05, 2.0, 0.1, P[None, :], Pidx, disc=disc, rhs=rid, eval_kernel=ek) as pr:
            kernels.add(pr.info()["eval_kernel"])
            A, me, fe, g = pr.action_grad(XP, rf)
        assert abs(A[0] - A0) <= 1e-12 * abs(A0), (ek, A[0], A0)
        assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max(), ek
    print("piecewise model D=%d N=%d %s ran eval kernels %s" % (D, N, disc, sorted(kernels)))


def test_bounded_step_one_lands_exactly_on_the_bound():
    """L-BFGS-B's `if (stp == 1) x = z`: with a bound far from the start point relative to its own size,
    x + (z - x) would land a few ulp inside it; the device takes z itself at step 1, so a variable the Cauchy point puts on a
    bound SITS on it, bit for bit, and the iteration counts follow the oracle's restatement of L-BFGS-B"""
    from varanneal_amd import twin
    D, N, B = 20, 200, 2
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    # tiny upper bounds on a third of the states (|u| << |x0 - u|), a binding box on the parameter
    bnds = [(-15.0, 1e-3 * (1 + (i % 7))) if i % 3 == 0 else (-15.0, 15.0) for i in range(N * D)] + [(6.5, 7.0)]
    lo = np.array([q[0] for q in bnds]); hi = np.array([q[1] for q in bnds])
    o = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 400, 'maxiter': 25}
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", bounds=bnds) as pb:
        r = pb.minimize_lbfgs(XP, 50.0, o)
    for b in range(B):
        x = r["x"][b]
        assert np.all(x >= lo) and np.all(x <= hi)
        on_hi = np.isclose(x, hi, rtol=0, atol=1e-9 * np.maximum(1.0, np.abs(hi)))
        assert on_hi.sum() > 100 and np.array_equal(x[on_hi], hi[on_hi])      # exactly on the bound, not a few ulp inside
        opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc="trapezoid")
        xo, Ao, sto, nito, nfevo = opb.minimize_lbfgs(XP[b], 50.0, o, bounds=bnds, exact=True)
        assert (r["nit"][b], r["nfev"][b], r["status"][b]) == (nito, nfevo, sto)
        assert np.array_equal(x == hi, xo == hi) and np.array_equal(x == lo, xo == lo)      # the same variables on their bounds
        assert abs(r["A"][b] - Ao) <= 1e-8 * abs(Ao)
