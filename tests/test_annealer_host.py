"""CPU: the host-side drop-in (varanneal_amd.va_ode.Annealer) -- argument handling,
ladder bookkeeping, result shapes and the reference's file formats.  The device is
replaced (in this test module only) by a stand-in built on the CPU oracle, so the
Annealer's own logic can be checked against the golden ladders the reference's
anneal() produced; the real device path is covered by tests/test_gpu_*.py."""
import os

import numpy as np
import pytest

import va_oracle
from varanneal_amd import _capi, rhs, twin, va_ode

OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


class OracleBackedProblem(object):
    """Test double for _capi.Problem (same methods/shapes), arithmetic by oracle/."""

    def __init__(self, batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, disc="trapezoid",
                 rhs="lorenz96", merr_nskip=1, **kw):
        P = np.asarray(P, dtype=np.float64).reshape(batch, -1)
        self.B, self.D, self.N, self.NP, self.NPest = batch, D, N_model, P.shape[1], len(Pidx)
        self.pbs = [va_oracle.Problem(D, N_model, Y, Lidx, dt_model, RM, RF0, P[b], Pidx, disc=disc,
                                      merr_nskip=merr_nskip) for b in range(batch)]
        self.kw = kw

    def close(self):
        pass

    def action_grad(self, XP, rf_scale=1.0, want_grad=True):
        r = [pb.action_grad(XP[b], rf_scale, want_grad) for b, pb in enumerate(self.pbs)]
        A, me, fe = (np.array([x[i] for x in r]) for i in range(3))
        return A, me, fe, (np.array([x[3] for x in r]) if want_grad else None)

    def minimize_lbfgs(self, XP, rf_scale, opt_args=None):
        out = dict(x=[], A=[], me=[], fe=[], status=[], nit=[], nfev=[])
        for b, pb in enumerate(self.pbs):
            x, A, st, nit, nfev = pb.minimize_lbfgs(XP[b], rf_scale, opt_args)
            _, me, fe, _ = pb.action_grad(x, rf_scale, want_grad=False)
            for k, v in zip(("x", "A", "me", "fe", "status", "nit", "nfev"), (x, A, me, fe, st, nit, nfev)):
                out[k].append(v)
        return {k: np.array(v) for k, v in out.items()}

    def anneal(self, XP, rf_scale, opt_args=None, want_paths=False, **kw):
        ND = self.N * self.D
        res = []
        for b, pb in enumerate(self.pbs):
            xp = np.array(XP[b]); rows = []
            for rf in rf_scale:
                x, A, st, nit, nfev = pb.minimize_lbfgs(xp, rf, opt_args)
                pb.P[pb.Pidx] = x[ND:]
                _, me, fe, _ = pb.action_grad(x, rf, want_grad=False)
                rows.append((A, me, fe, st, nit, nfev, np.append(x[:ND], pb.P), x[ND:].copy()))
                xp = x
            res.append(rows)
        g = lambda i, dt=np.float64: np.array([[r[i] for r in rows] for rows in res], dtype=dt)
        return dict(x=None, A=g(0), me=g(1), fe=g(2), status=g(3, np.int32), nit=g(4, np.int32),
                    nfev=g(5, np.int64), minpaths=g(6), pest=g(7))


@pytest.fixture
def fake_device(monkeypatch):
    monkeypatch.setattr(_capi, "Problem", OracleBackedProblem)


def _setup(c):
    a = va_ode.Annealer()
    a.set_model(twin.l96, int(c["D"]))
    a.set_data(c["Y"], t=c["t"])
    return a


def test_rhs_recognition():
    assert rhs.recognise(twin.l96, 20) == "lorenz96"
    assert rhs.recognise("lorenz96", 20) == "lorenz96"
    assert rhs.recognise(lambda t, x, p: -x, 20) is None
    assert rhs.recognise(lambda t, x, p: np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - x + 2 * p, 8) is None

    def loop_l96(t, x, k):            # the tutorial's loop form (VarAnneal_tutorial.ipynb:117-131)
        D = x.shape[1]
        out = np.empty_like(x)
        for i in range(D):
            out[:, i] = x[:, (i - 1) % D] * (x[:, (i + 1) % D] - x[:, (i - 2) % D]) - x[:, i] + k
        return out
    assert rhs.recognise(loop_l96, 5) == "lorenz96"


def test_set_data_variants():
    t, Y, _, Lidx = twin.make_twin(20, 50)
    a = va_ode.Annealer()
    a.set_data(np.column_stack([t, Y]))              # time in column 0 (va_ode.py:109-116)
    assert a.N_data == 50 and np.allclose(a.Y, Y) and abs(a.dt_data - twin.DT) < 1e-15
    a.set_data(Y, t=t, nstart=5, N=20)
    assert a.N_data == 20 and np.array_equal(a.Y, Y[5:25]) and a.t_data[0] == t[5]


def test_set_data_fromfile_works(tmp_path):
    t, Y, _, Lidx = twin.make_twin(20, 30)
    f = tmp_path / "d.npy"
    np.save(f, np.column_stack([t, Y]))
    a = va_ode.Annealer()
    a.set_data_fromfile(str(f))                      # upstream raises NameError here (va_ode.py:96)
    assert a.N_data == 30 and np.allclose(a.Y, Y)


def test_unsupported_inputs_fail_loudly(fake_device, golden_ladders):
    c = golden_ladders["g4_c1_trapezoid_N200"]
    args = lambda: (c["X0"].copy(), c["P0"].copy(), 1.5, np.arange(3), 4.0, 4e-6, list(c["Lidx"]), [0])
    a = _setup(c)
    a.set_model(lambda t, x, p: np.where(x > 0, x, -x) if x.dtype != object else (x if x[0, 0] > 0 else -x), 20)
    with pytest.raises(TypeError):                   # a model that branches on its inputs cannot be traced
        a.anneal_init(*args())
    a = _setup(c)
    with pytest.raises(NotImplementedError):         # time-dependent parameters: not with euler (broken upstream)
        a.anneal_init(c["X0"].copy(), np.ones((200, 1)), 1.5, np.arange(3), 4.0, 4e-6, list(c["Lidx"]), [0],
                      disc="euler")
    with pytest.raises(ValueError):                  # ... and one row per model time point
        a.anneal_init(c["X0"].copy(), np.ones((199, 1)), 1.5, np.arange(3), 4.0, 4e-6, list(c["Lidx"]), [0])
    # (full RF matrices, (D, D) / (N-1, D, D), are accepted since round 2: tests/test_rffull.py)
    with pytest.raises(ValueError):
        a.anneal_init(c["X0"].copy(), c["P0"].copy(), 1.5, np.arange(3), 4.0, np.ones(3), list(c["Lidx"]), [0])
    with pytest.raises(ValueError):
        a.anneal_init(*args(), disc="rk4")
    assert a.anneal_init(*args(), method="BFGS") is None     # reference prints and returns None


def test_ladder_through_the_annealer_matches_reference(fake_device, golden_ladders, tmp_path):
    c = golden_ladders["g4_c1_trapezoid_N200"]
    N, D, nb = int(c["N"]), int(c["D"]), len(c["beta"])
    a = _setup(c)
    X0 = c["X0"].copy()
    a.anneal(X0, c["P0"].copy(), float(c["alpha"]), c["beta"], 4.0, 4e-6, list(c["Lidx"]), [0],
             dt_model=float(c["t"][1] - c["t"][0]), init_to_data=True, disc="trapezoid",
             method="L-BFGS-B", opt_args=OPTS, adolcID=0, verbose=False)
    # reference quirk kept: init_to_data writes the data into the caller's X0 (va_ode.py:677-678)
    assert np.array_equal(X0[:, c["Lidx"]], c["Y"])
    assert a.minpaths.shape == (nb, N * D + 1) and a.A_array.shape == (nb,)
    assert np.all(np.abs(a.A_array[:12] - c["A_array"][:12]) <= 1e-8)
    assert abs(a.A_array[-1] - c["A_array"][-1]) <= 1e-3 * c["A_array"][-1]
    assert abs(a.P[0] - c["params"][-1, 0]) <= 2e-3 * abs(c["params"][-1, 0])
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    assert a.betaidx == nb - 1 and a.beta == c["beta"][-1]
    # file formats (va_ode.py:794-873)
    a.save_paths(str(tmp_path / "paths.npy")); a.save_params(str(tmp_path / "params.npy"))
    a.save_action_errors(str(tmp_path / "ae.npy"))
    paths, params, ae = (np.load(str(tmp_path / f)) for f in ("paths.npy", "params.npy", "ae.npy"))
    assert paths.shape == (nb, N, D + 1) and np.array_equal(paths[3, :, 0], a.t_model)
    assert np.array_equal(paths[-1, :, 1:].ravel(), a.minpaths[-1, :N * D])
    assert params.shape == (nb, 1) and params[-1, 0] == a.P[0]
    assert ae.shape == (nb, 5) and np.array_equal(ae[:, 0], c["beta"]) and np.array_equal(ae[:, 1], a.A_array)
    assert np.allclose(ae[:, 4], a.fe_array / (4e-6 * 1.5 ** c["beta"]))
    a.save_as_minAone(str(tmp_path))
    txt = np.loadtxt(str(tmp_path / "D20_M7_PATH0.dat"))
    assert txt.shape == (nb, 3 + N * D + 1)
    a.save_action_errors(str(tmp_path / "ae.txt"))
    assert np.loadtxt(str(tmp_path / "ae.txt")).shape == (nb, 5)


def test_stepwise_equals_fused_and_tracking(fake_device, golden_ladders, tmp_path):
    c = golden_ladders["g4_shipped_SH_N161"]
    nb = 8
    run = {}
    for mode in ("fused", "steps"):
        a = _setup(c)
        kw = {}
        if mode == "steps":
            kw = dict(track_paths={"filename": str(tmp_path / "tp.npy")},
                      track_action_errors={"filename": str(tmp_path / "tae.txt"), "fmt": "%.10e"})
        a.anneal(c["X0"].copy(), c["P0"].copy(), 1.5, np.arange(nb), 4.0, 4e-6, list(c["Lidx"]), [0],
                 disc="SimpsonHermite", opt_args=OPTS, verbose=False, **kw)
        run[mode] = a
    assert np.allclose(run["fused"].A_array, run["steps"].A_array, rtol=1e-12)
    assert np.allclose(run["fused"].minpaths, run["steps"].minpaths, rtol=0, atol=1e-12)
    assert np.all(np.abs(run["steps"].A_array - c["A_array"][:nb]) <= 1e-8)
    assert os.path.exists(str(tmp_path / "tp.npy")) and np.loadtxt(str(tmp_path / "tae.txt")).shape == (nb, 5)


def test_batched_seeds_and_vector_weights(fake_device):
    D, N, B, nb = 20, 41, 3, 4
    t, Y, _, Lidx = twin.make_twin(D, N)
    X0 = np.empty((B, N, D)); P0 = np.empty((B, 1))
    for b in range(B):
        X0[b], P0[b] = twin.initial_guess(N, D, b)
    a = va_ode.Annealer()
    a.set_model("lorenz96", D)
    a.set_data(Y, t=t)
    a.anneal(X0, P0, 1.5, np.arange(nb), [4.0] * len(Lidx), list(4e-6 * np.ones(D)), Lidx, [0],
             disc="SimpsonHermite", opt_args=OPTS, verbose=False)
    assert a.minpaths.shape == (B, nb, N * D + 1) and a.A_array.shape == (B, nb)
    # each seed equals its own single-seed run
    b = 2
    s = va_ode.Annealer(); s.set_model("lorenz96", D); s.set_data(Y, t=t)
    Xs, Ps = twin.initial_guess(N, D, b)
    s.anneal(Xs, Ps, 1.5, np.arange(nb), 4.0, 4e-6, Lidx, [0], disc="SimpsonHermite", opt_args=OPTS, verbose=False)
    assert np.allclose(a.A_array[b], s.A_array, rtol=1e-10)
    assert a.P.shape == (B, 1) and np.allclose(a.P[b], s.P)


def test_scipy_route_with_bounds(fake_device, golden_ladders):
    c = golden_ladders["g4_c1_trapezoid_N200"]
    a = _setup(c)
    bounds = [(-15.0, 15.0)] * 20 + [(6.5, 10.0)]
    a.anneal(c["X0"].copy(), c["P0"].copy(), 1.5, np.arange(20), 4.0, 4e-6, list(c["Lidx"]), [0],
             opt_args=OPTS, bounds=bounds, verbose=False, bounded_minimiser='scipy')
    assert len(a.bounds) == 200 * 20 + 1 and a.bounds[-1] == (6.5, 10.0)
    assert np.all(a.minpaths[:, -1] >= 6.5 - 1e-12) and np.all(np.abs(a.minpaths[:, :-1]) <= 15.0 + 1e-12)
    assert np.all(a.A_array > 0) and np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    assert np.all(a.exitflags == 0)


class EmulBackedProblem(object):
    """Test double for _capi.Problem on a GENERATED right-hand side: single evaluations by the
    CPU emulator compiled with the generated header (same tile phases as the device module)."""
    headers = {}

    def __init__(self, batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, disc="trapezoid",
                 rhs=0, merr_nskip=1, t_model=None, stim=None, **kw):
        from cpu_emul import emul
        self.emul, self.B = emul, batch
        self.header = self.headers[rhs]
        self.desc, self.keep = _capi.make_desc(batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, disc=disc,
                                               rhs=1000, merr_nskip=merr_nskip, t_model=t_model, stim=stim)

    def close(self):
        pass

    def action_grad(self, XP, rf_scale=1.0, want_grad=True):
        A, me, fe, g = self.emul.action_grad(self.desc, 16, np.asarray(XP), rf_scale, user_header=self.header)
        return A, me, fe, (g if want_grad else None)


def test_nakl_bounded_ladder_host_flow(monkeypatch):
    """Stimulus + bounds + generated RHS through the Annealer (tutorial NaKL flow,
    va_ode.py:582-605 bounds expansion, :356 f(t, x, (p, stim))) against the ladder the
    reference produced (tests/golden/nakl.npz)."""
    from _util import load_npz_cases
    from models.nakl import nakl
    c = load_npz_cases("nakl.npz")["g5_nakl_ladder_SH_N101"]

    from varanneal_amd import codegen
    built, real_module_for = {}, codegen.module_for

    def recording_module_for(*a, **k):                      # remember the generated header of the module just built
        built["last"] = real_module_for(*a, **k)
        return built["last"]

    def fake_load(path):
        assert path == built["last"]["so"]
        EmulBackedProblem.headers[1000 + len(EmulBackedProblem.headers)] = built["last"]["header"]
        return 1000 + len(EmulBackedProblem.headers) - 1
    monkeypatch.setattr(codegen, "module_for", recording_module_for)
    monkeypatch.setattr(_capi, "load_rhs_module", fake_load)
    monkeypatch.setattr(_capi, "Problem", EmulBackedProblem)
    a = va_ode.Annealer()
    a.set_model(nakl, 4)
    a.set_data(c["Y"], stim=c["stim"], t=c["t"])
    a.anneal(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), c["beta"], 1.0, list(c["RF0"]), [0],
             list(range(18)), dt_model=None, init_to_data=True, disc="SimpsonHermite", method='L-BFGS-B',
             bounds=[tuple(b) for b in c["bounds"]], opt_args=OPTS, adolcID=0, verbose=False, bounded_minimiser='scipy')
    assert list(a.nit_array[:4]) == list(c["nit"][:4])
    assert np.all(np.abs(a.A_array[:7] - c["A_array"][:7]) <= 1e-5 * c["A_array"][:7])
    assert np.all(np.abs(a.A_array - c["A_array"]) <= 2e-2 * c["A_array"])
    lo, hi = c["bounds"][4:, 0], c["bounds"][4:, 1]
    assert np.all(a.P >= lo - 1e-12) and np.all(a.P <= hi + 1e-12)


class EmulTdpProblem(object):
    """Test double for _capi.Problem with time-dependent parameters on the built-in Lorenz-96:
    evaluations by the CPU emulator's flat tile phases."""

    def __init__(self, batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, disc="trapezoid",
                 rhs="lorenz96", merr_nskip=1, t_model=None, stim=None, p_time_dependent=False, **kw):
        from cpu_emul import emul
        assert p_time_dependent and rhs == "lorenz96"
        self.emul, self.B = emul, batch
        self.desc, self.keep = _capi.make_desc(batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, disc=disc,
                                               merr_nskip=merr_nskip, t_model=t_model, p_time_dependent=True)

    def close(self):
        pass

    def action_grad(self, XP, rf_scale=1.0, want_grad=True):
        A, me, fe, g = self.emul.action_grad(self.desc, 9, np.asarray(XP), rf_scale)
        return A, me, fe, (g if want_grad else None)


def test_time_dependent_parameters_host_flow(monkeypatch, tmp_path):
    """P0 of shape (N_model, NP) (va_ode.py:565-570): packing [X | P time-major], warm starts,
    write-back into the caller's P, bounds repeated per time point, save_params (Nbeta, N, NP) --
    against the ladder the reference produced (tests/golden/tdp.npz; SciPy route here because the
    stand-in only evaluates)."""
    from _util import load_npz_cases
    c = load_npz_cases("tdp.npz")["g8_tdp_ladder_SH_N41"]
    monkeypatch.setattr(_capi, "Problem", EmulTdpProblem)
    N, D, nb = int(c["N"]), int(c["D"]), 5
    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(c["Y"], t=c["t"])
    X0, P0 = c["X0"].copy(), c["P0"].copy()
    # open bounds: L-BFGS-B then takes the same steps as unbounded (any finite bound, even an
    # inactive one, changes its first step length), while the expansion over time points is exercised
    big = [(None, None)] * (D + 1)
    a.anneal(X0, P0, float(c["alpha"]), c["beta"][:nb], 4.0, 4e-6, list(c["Lidx"]), [0], dt_model=None,
             init_to_data=True, disc="SimpsonHermite", method='L-BFGS-B', bounds=big, opt_args=OPTS, adolcID=0,
             verbose=False, bounded_minimiser='scipy')
    assert len(a.bounds) == N * D + N
    assert list(a.nit_array) == list(c["nit"][:nb])
    assert np.all(np.abs(a.A_array - c["A_array"][:nb]) <= 1e-6 * c["A_array"][:nb])
    assert a.minpaths.shape == (nb, N * D + N) and P0.shape == (N, 1)
    assert np.array_equal(P0[:, 0], a.minpaths[-1, N * D:])           # written back into the caller's array
    a.save_params(str(tmp_path / "p.npy"))
    assert np.load(str(tmp_path / "p.npy")).shape == (nb, N, 1)
    with pytest.raises(NotImplementedError):
        a.anneal(X0, P0, 1.5, np.arange(2), 4.0, 4e-6, list(c["Lidx"]), [0], disc="euler", verbose=False)


def test_admin_surface(fake_device, golden_ladders):
    """the ADmin method names user code may call directly (_autodiffmin.py:32-143)"""
    c = golden_ladders["g4_c1_trapezoid_N200"]
    a = _setup(c)
    a.anneal_init(c["X0"].copy(), c["P0"].copy(), 1.5, np.arange(3), 4.0, 4e-6, list(c["Lidx"]), [0],
                  opt_args=OPTS, verbose=False)
    XP0 = a._xp0(0)[0]
    a.tape_A(a.gen_xtrace())
    assert a.gen_xtrace().shape == XP0.shape
    A, g = a.A_gradA_taped(XP0)
    assert a.A_taped(XP0) == A and np.array_equal(a.gradA_taped(XP0), g)
    x, Amin, status = a.min_lbfgs_scipy(XP0)
    assert x.shape == XP0.shape and Amin < A and status in (0, 1, 2)
    x2, A2, st2 = a.min_cg_scipy(XP0)
    x3, A3, st3 = a.min_tnc_scipy(XP0)
    assert A2 < A and A3 < A
    with pytest.raises(NotImplementedError):
        a.hessianA_taped(XP0)
    with pytest.raises(NotImplementedError):
        a.min_lm_scipy(XP0)


def test_integer_alpha_with_a_long_ladder(fake_device, golden_ladders, tmp_path):
    """RF = RF0 * alpha**beta in float64 whatever the types: beta_array is uint16 (va_ode.py:644) and
    NumPy 2 would evaluate 2 ** uint16(16) in uint16 (= 0), switching the model term off."""
    from varanneal_amd._hipmin import alpha_pow
    assert np.array_equal(alpha_pow(2, np.array([0, 15, 16, 30], np.uint16)), [1.0, 2.0 ** 15, 2.0 ** 16, 2.0 ** 30])
    c = golden_ladders["g4_c1_trapezoid_N200"]
    a = _setup(c)
    beta = np.array([0, 16, 30])
    a.anneal(c["X0"].copy(), c["P0"].copy(), 2, beta, 4.0, 4e-6, list(c["Lidx"]), [0],
             dt_model=float(c["t"][1] - c["t"][0]), init_to_data=True, disc="trapezoid",
             method="L-BFGS-B", opt_args=dict(OPTS, maxiter=3), adolcID=0, verbose=False)
    assert np.array_equal(a._rf_scale, [1.0, 2.0 ** 16, 2.0 ** 30]) and a.RF == 4e-6 * 2.0 ** 30
    assert np.all(a.fe_array > 0.0)
    a.save_action_errors(str(tmp_path / "ae.npy"))
    ae = np.load(str(tmp_path / "ae.npy"))
    assert np.all(np.isfinite(ae)) and np.allclose(ae[:, 4], a.fe_array / (4e-6 * 2.0 ** beta))
