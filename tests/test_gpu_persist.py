"""GPU tests of the persistent per-seed ladder kernel (csrc/va_persist.h): ONE launch runs the whole RF
ladder of a seed, every vector of the minimisation resident in the LDS of its workgroups -- the path the drop-in takes
for the reference's default use (examples/Lorenz96_D20/Lorenz96_anneal.py:84-86: one seed, N = 161).

Checked against the CPU oracle (same optimiser: identical (nit, nfev, status), iterates to 1e-6) and against the
three-launch cycle of the same library (`tune persist=0`); the two device paths add their partial sums in different
orders, so they agree to rounding, not bit for bit, and long ladders may leave each other where the landscape
bifurcates (as device and oracle do: tests/test_gpu_parity.py).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


@pytest.fixture(scope="module")
def capi():
    from varanneal_amd import _capi
    _capi.lib()
    return _capi


def twin_problem(D, N, B, nskip=1):
    from varanneal_amd import twin
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    return (Y[::nskip] if nskip > 1 else Y), Lidx, XP, P, twin.DT


def test_chosen_for_few_seeds_only(capi):
    Y, Lidx, XP, P, dt = twin_problem(20, 200, 1)
    with capi.Problem(1, 20, 200, Y, Lidx, dt, 4.0, 4e-6, P, [0]) as pb:
        geo = pb.persistent()
        assert geo is not None and geo[0] * geo[1] >= 200 and (geo[0] - 1) * geo[1] < 200
        pb.tune(persist=0)
        assert pb.persistent() is None
        pb.tune(persist=1)
        with pytest.raises(capi.VaError):
            pb.tune(persist_rows=1)                     # a slice needs two rows
        with pytest.raises(capi.VaError):
            pb.tune(persist_rows=150)                   # does not fit the LDS
    # 64 seeds of N = 1000 need more workgroups than the chip has CUs: three-launch cycle
    Y, Lidx, XP, P, dt = twin_problem(20, 1000, 64)
    with capi.Problem(64, 20, 1000, Y, Lidx, dt, 4.0, 4e-6, P, [0]) as pb:
        assert pb.persistent() is None
    # bounds: L-BFGS-B's direction step is a kernel of its own
    Y, Lidx, XP, P, dt = twin_problem(20, 200, 1)
    with capi.Problem(1, 20, 200, Y, Lidx, dt, 4.0, 4e-6, P, [0], bounds=[(-15.0, 15.0)] * (200 * 20 + 1)) as pb:
        assert pb.persistent() is None


@pytest.mark.parametrize("disc,N,rf,maxiter", [("trapezoid", 200, 1.0, 1000000), ("trapezoid", 200, 1.5 ** 15, 25),
                                               ("SimpsonHermite", 201, 1.5 ** 7, 40), ("euler", 200, 1.5 ** 10, 30),
                                               ("forwardmap", 200, 1.0, 12), ("trapezoid", 1000, 1.5 ** 12, 20)])
def test_minimisation_matches_oracle_step_for_step(capi, disc, N, rf, maxiter):
    import va_oracle
    D = 20
    Y, Lidx, XP, P, dt = twin_problem(D, N, 2)
    o = dict(OPTS, maxiter=maxiter)
    with capi.Problem(2, D, N, Y, Lidx, dt, 4.0, 4e-6, P, [0], disc=disc) as pb:
        assert pb.persistent() is not None
        r = pb.minimize_lbfgs(XP, rf, o)
        pb.tune(persist=0)
        r3 = pb.minimize_lbfgs(XP, rf, o)
    for b in range(2):
        opb = va_oracle.Problem(D, N, Y, Lidx, dt, 4.0, 4e-6, P[b], [0], disc=disc)
        x, A, st, nit, nfev = opb.minimize_lbfgs(XP[b], rf, o)
        assert (r["nit"][b], r["nfev"][b], r["status"][b]) == (nit, nfev, st), (disc, b, r["nit"][b], nit)
        assert (r3["nit"][b], r3["nfev"][b], r3["status"][b]) == (nit, nfev, st)
        assert abs(r["A"][b] - A) <= 1e-6 * abs(A)
        assert np.abs(r["x"][b] - x).max() <= 1e-6 * max(1.0, np.abs(x).max())
        assert abs(r["A"][b] - (r["me"][b] + r["fe"][b])) <= 1e-12 * abs(A)


@pytest.mark.parametrize("variant", ["rf_vec", "rm_vec", "nskip2"])
def test_weights_and_sparse_data(capi, variant):
    import va_oracle
    D, N = 20, 201 if variant == "nskip2" else 200
    nskip = 2 if variant == "nskip2" else 1
    Y, Lidx, XP, P, dt = twin_problem(D, N, 1, nskip)
    RM, RF0 = 4.0, 4e-6
    if variant == "rf_vec":
        RF0 = np.resize(4e-6 * (1.0 + 0.1 * np.arange(D)), (N - 1, D))
    if variant == "rm_vec":
        RM = np.resize(4.0 * (1.0 + 0.1 * np.arange(len(Lidx))), Y.shape)
    o = dict(OPTS, maxiter=30)
    with capi.Problem(1, D, N, Y, Lidx, dt, RM, RF0, P, [0], merr_nskip=nskip) as pb:
        assert pb.persistent() is not None
        r = pb.minimize_lbfgs(XP, 1.5 ** 9, o)
        pb.tune(persist=0)                    # the three-launch cycle: k_eval4's variants with weight arrays / the row mask of merr_nskip
        r3 = pb.minimize_lbfgs(XP, 1.5 ** 9, o)
    opb = va_oracle.Problem(D, N, Y, Lidx, dt, RM, RF0, P[0], [0], merr_nskip=nskip)
    x, A, st, nit, nfev = opb.minimize_lbfgs(XP[0], 1.5 ** 9, o)
    for rr in (r, r3):
        assert (rr["nit"][0], rr["nfev"][0], rr["status"][0]) == (nit, nfev, st)
        assert abs(rr["A"][0] - A) <= 1e-6 * abs(A) and np.abs(rr["x"][0] - x).max() <= 1e-6 * np.abs(x).max()


def test_slice_sizes_agree(capi):
    """the same minimisation with 8, 14 and the default number of rows per workgroup (25, 15 and 7 workgroups): the
    slices change only the order of the partial sums"""
    D, N = 20, 200
    Y, Lidx, XP, P, dt = twin_problem(D, N, 1)
    o = dict(OPTS, maxiter=30)
    out = []
    with capi.Problem(1, D, N, Y, Lidx, dt, 4.0, 4e-6, P, [0]) as pb:
        for rows in (0, 8, 14):
            pb.tune(persist_rows=rows)
            geo = pb.persistent()
            assert rows == 0 or geo == ((N + rows - 1) // rows, rows)
            out.append(pb.minimize_lbfgs(XP, 1.5 ** 8, o))
    for r in out[1:]:
        assert (r["nit"][0], r["nfev"][0], r["status"][0]) == (out[0]["nit"][0], out[0]["nfev"][0], out[0]["status"][0])
        assert np.abs(r["x"][0] - out[0]["x"][0]).max() <= 1e-8 * np.abs(out[0]["x"][0]).max()


def test_ladder_against_three_launch_cycle_and_outputs(capi):
    """a 30-rung ladder: the bottom of the ladder rung for rung equal to the three-launch cycle's; every stored
    minimiser has its stored action; seeds of a batch are independent of their companions, bit for bit"""
    D, N, nb = 20, 200, 30
    Y, Lidx, XP, P, dt = twin_problem(D, N, 3)
    rf = 1.5 ** np.arange(nb)
    with capi.Problem(3, D, N, Y, Lidx, dt, 4.0, 4e-6, P, [0], max_beta=nb, keep_paths=1) as pb:
        assert pb.persistent() is not None
        r = pb.anneal(XP, rf, OPTS, want_paths=True)
        c_p = pb.counters()
        pb.tune(persist=0)
        r3 = pb.anneal(XP, rf, OPTS, want_paths=True)
        pb.tune(persist=1)
        for k in (0, nb // 2, nb - 1):
            A, me, fe, _ = pb.action_grad(np.concatenate([r["minpaths"][:, k, :N * D], r["pest"][:, k]], axis=1), rf[k], want_grad=False)
            assert np.all(np.abs(A - r["A"][:, k]) <= 1e-12 * np.abs(A))
    assert c_p["cycles"] > 0 and c_p["seed_evals"] == int(r["nfev"].sum())
    assert np.array_equal(r["nit"][:, :10], r3["nit"][:, :10]) and np.array_equal(r["nfev"][:, :10], r3["nfev"][:, :10])
    assert np.all(np.abs(r["A"][:, :10] - r3["A"][:, :10]) <= 1e-7 * np.abs(r3["A"][:, :10]))      # (both stop within ftol = 1e-8 of their minima)
    assert np.all(r["status"] == 0)
    assert np.all(np.abs(r["A"][:, -1] - r3["A"][:, -1]) <= 1e-3 * np.abs(r3["A"][:, -1]))
    assert np.array_equal(r["minpaths"][:, -1, :N * D], r["x"][:, :N * D])
    assert np.array_equal(r["minpaths"][:, :, N * D], r["pest"][:, :, 0])
    # seed 1 alone
    with capi.Problem(1, D, N, Y, Lidx, dt, 4.0, 4e-6, P[1:2], [0], max_beta=nb, keep_paths=1) as pb1:
        r1 = pb1.anneal(XP[1:2], rf, OPTS, want_paths=True)
    assert np.array_equal(r1["x"][0], r["x"][1]) and np.array_equal(r1["A"][0], r["A"][1]) and np.array_equal(r1["nfev"][0], r["nfev"][1])


def test_generated_model_runs_the_persistent_kernel(capi):
    """a traced user model (damped Lorenz-96, two parameters, one estimated) through its module's k_seed instantiation"""
    from varanneal_amd import codegen, twin

    def damped(t, x, p):
        return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[1] * x + p[0]
    D, N = 8, 300
    mod = codegen.module_for(damped, D, 2)
    rhs = capi.load_rhs_module(mod["so"])
    t, Y, _, Lidx = twin.make_twin(D, N)
    rng = np.random.RandomState(5)
    X0 = 20.0 * rng.rand(N, D) - 10.0
    X0[:, Lidx] = Y
    P = np.array([[7.0, 1.0]])
    XP = np.append(X0.ravel(), 7.0)[None, :]
    o = dict(OPTS, maxiter=40)
    with capi.Problem(1, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], rhs=rhs) as pb:
        assert pb.persistent() is not None
        r = pb.minimize_lbfgs(XP, 1.5 ** 6, o)
        pb.tune(persist=0)
        r3 = pb.minimize_lbfgs(XP, 1.5 ** 6, o)
    assert (r["nit"][0], r["nfev"][0], r["status"][0]) == (r3["nit"][0], r3["nfev"][0], r3["status"][0])
    assert abs(r["A"][0] - r3["A"][0]) <= 1e-8 * abs(r3["A"][0])
    assert np.abs(r["x"][0] - r3["x"][0]).max() <= 1e-6 * np.abs(r3["x"][0]).max()
