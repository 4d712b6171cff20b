"""GPU parity of the streaming evaluation kernel k_eval5 (csrc/va_eval5.h: column strips that march along
time; the production kernel of BASELINE config 4, Lorenz-96 D = 200) against the CPU oracle
(oracle/va_oracle.c, pinned by the reference's goldens) and against the tile kernel k_eval3.

Tolerances as tests/test_gpu_parity.py: |A-A_o|/|A_o| <= 1e-12, ||g-g_o||_inf/||g_o||_inf <= 1e-10."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_A = 1e-12
RTOL_G = 1e-10
OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


@pytest.fixture(scope="module")
def capi():
    from varanneal_amd import _capi
    _capi.lib()
    return _capi


def make(D, N, B, seed, L=None):
    from varanneal_amd import twin
    t, Y, _, Lidx = twin.make_twin(D, N)
    rng = np.random.RandomState(seed)
    if L is not None:
        Lidx = np.sort(rng.choice(D, L, replace=False))
        Y = rng.randn(N, L)
    XP = np.concatenate([3.0 * rng.randn(B, N * D), 6.0 + 3.0 * rng.rand(B, 1)], axis=1)
    return Y, list(Lidx), XP, XP[:, -1:].copy()


def check_vs_oracle(capi, D, N, B, disc, tile_rows, seed=1, L=None, rf=37.0):
    import va_oracle
    from varanneal_amd import twin
    Y, Lidx, XP, P = make(D, N, B, seed, L)
    with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc=disc, eval_kernel=5, tile_rows=tile_rows) as pb:
        info = pb.info()
        assert info["eval_kernel"] == 5, info
        A, me, fe, g = pb.action_grad(XP, rf)
        A2, me2, fe2, g2 = pb.action_grad(XP, rf)
        assert np.array_equal(A, A2) and np.array_equal(g, g2) and np.array_equal(me, me2) and np.array_equal(fe, fe2)
    for b in range(B):
        opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc=disc)
        Ao, meo, feo, go = opb.action_grad(XP[b], rf)
        assert abs(A[b] - Ao) <= RTOL_A * abs(Ao), (D, N, b, disc, tile_rows, A[b], Ao)
        assert abs(me[b] - meo) <= RTOL_A * max(abs(meo), abs(Ao)), (D, N, b, disc, tile_rows)
        assert abs(fe[b] - feo) <= RTOL_A * abs(feo), (D, N, b, disc, tile_rows)
        err = np.abs(g[b] - go)
        assert err.max() <= RTOL_G * np.abs(go).max(), (D, N, b, disc, tile_rows, int(err.argmax()), err.max())
    return info


@pytest.mark.parametrize("disc", ["trapezoid", "euler", "forwardmap", "SimpsonHermite"])
def test_c4_shape_small(capi, disc):
    """D = 200 (the compiled-in width): one segment, several segments, segments of odd and even stream length
    (Simpson-Hermite: an odd number of rows, va_ode.py:231-233 upstream; a last segment of one row)"""
    odd = disc == "SimpsonHermite"
    for N, B, tile_rows in ((40, 3, 0), (301, 2, 0), (300, 2, 50), (97, 4, 32), (64, 1, 33), (129, 2, 32), (67, 1, 64)):
        check_vs_oracle(capi, 200, N + (odd and N % 2 == 0), B, disc, tile_rows, seed=N)


def test_other_widths(capi):
    """run-time D: two strips (D = 66, 100), a last strip narrower than the others (D = 130: 44 + 44 + 42),
    more strips than one workgroup holds (D = 500: 9 strips, three workgroups per row)"""
    for D, N, B in ((66, 120, 2), (100, 90, 3), (130, 75, 2), (500, 60, 2), (256, 50, 1)):
        L = 2 * (D // 5)
        info = check_vs_oracle(capi, D, N, B, "trapezoid", 0, seed=D, L=L)
        check_vs_oracle(capi, D, N, B, "trapezoid", 34, seed=D + 1, L=L)
        check_vs_oracle(capi, D, N + 1 - N % 2, B, "SimpsonHermite", 34, seed=D + 2, L=L)
        print(D, info)


def test_sparse_and_dense_observations(capi):
    """strips without any observed column, every column observed, observed columns in one strip only"""
    import va_oracle
    from varanneal_amd import twin
    D, N, B = 200, 70, 2
    for Lidx in ([0, 1], list(range(0, 200)), [150, 151, 152, 199], list(range(3, 200, 2))[:98], [77]):
        rng = np.random.RandomState(len(Lidx))
        Y = rng.randn(N, len(Lidx))
        XP = np.concatenate([2.0 * rng.randn(B, N * D), 7.0 + rng.rand(B, 1)], axis=1)
        P = XP[:, -1:].copy()
        with capi.Problem(B, D, N, Y, Lidx, twin.DT, 3.0, 0.7, P, [0], disc="trapezoid", eval_kernel=5, tile_rows=36) as pb:
            assert pb.info()["eval_kernel"] == 5
            A, me, fe, g = pb.action_grad(XP, 2.5)
        for b in range(B):
            opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 3.0, 0.7, P[b], [0], disc="trapezoid")
            Ao, meo, feo, go = opb.action_grad(XP[b], 2.5)
            assert abs(me[b] - meo) <= RTOL_A * abs(meo), (len(Lidx), b)
            assert abs(A[b] - Ao) <= RTOL_A * abs(Ao)
            assert np.abs(g[b] - go).max() <= RTOL_G * np.abs(go).max()


def test_fallback_when_the_streaming_kernel_does_not_apply(capi):
    """odd D (rows that are not 16-byte aligned), full weight matrices: the handle runs another kernel instead (and
    says so); Simpson-Hermite, per-row weights and an odd number of observed columns stream"""
    from varanneal_amd import twin
    D, N, B = 200, 41, 1
    Y, Lidx, XP, P = make(201, N, B, 3, L=8)
    with capi.Problem(B, 201, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", eval_kernel=5) as pb:
        assert pb.info()["eval_kernel"] == 3
    Y, Lidx, XP, P = make(D, N, B, 3)
    with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="SimpsonHermite", eval_kernel=5) as pb:
        assert pb.info()["eval_kernel"] == 5
    with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, np.full((N - 1, D), 4e-6), P, [0], disc="trapezoid", eval_kernel=5) as pb:
        assert pb.info()["eval_kernel"] == 5
    with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, np.full((N - 1, D, D), 4e-6), P, [0], disc="trapezoid", eval_kernel=5) as pb:
        assert pb.info()["eval_kernel"] == 1


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite"])
def test_odd_number_of_observed_columns(capi, disc):
    """L odd: the data rows get a pad column on the device (they are staged by 16-byte pieces); with and without
    per-row measurement weights"""
    import va_oracle
    from varanneal_amd import twin
    for D, N, L in ((200, 61, 7), (200, 45, 41), (130, 51, 1)):
        check_vs_oracle(capi, D, N, 2, disc, 30, seed=L, L=L)
    D, N, B, L = 200, 61, 2, 9
    rng = np.random.RandomState(9)
    Lidx = np.sort(rng.choice(D, L, replace=False))
    Y = rng.randn(N, L)
    RM = 0.5 + rng.rand(N, L)
    XP = np.concatenate([2.0 * rng.randn(B, N * D), 7.0 + rng.rand(B, 1)], axis=1)
    P = XP[:, -1:].copy()
    with capi.Problem(B, D, N, Y, list(Lidx), twin.DT, RM, 0.7, P, [0], disc=disc, eval_kernel=5, tile_rows=24) as pb:
        assert pb.info()["eval_kernel"] == 5
        A, me, fe, g = pb.action_grad(XP, 2.5)
    for b in range(B):
        Ao, meo, feo, go = va_oracle.Problem(D, N, Y, list(Lidx), twin.DT, RM, 0.7, P[b], [0], disc=disc).action_grad(XP[b], 2.5)
        assert abs(A[b] - Ao) <= RTOL_A * abs(Ao) and abs(me[b] - meo) <= RTOL_A * max(abs(meo), abs(Ao))
        assert np.abs(g[b] - go).max() <= RTOL_G * np.abs(go).max()


def test_timed_evaluation_is_a_complete_evaluation(capi):
    from varanneal_amd import twin
    D, N, B = 200, 500, 9
    Y, Lidx, XP, P = make(D, N, B, 5)
    with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", eval_kernel=5) as pb:
        A, me, fe, g = pb.action_grad(XP, 11.0)
        for iters in (1, 7):
            pb.eval_timed(11.0, iters)
            A2, me2, fe2, g2 = pb.read_eval_outputs()
            assert np.array_equal(A, A2) and np.array_equal(me, me2) and np.array_equal(fe, fe2) and np.array_equal(g, g2)


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite"])
def test_minimisation_follows_the_oracle(capi, disc):
    """line-search evaluations (x + stp d formed in LDS from the two staged images): a short minimisation takes
    the oracle's (nit, nfev, status) and ends at its point; the tile kernel gives the same counts"""
    import va_oracle
    from varanneal_amd import twin
    D, N, B = 200, 121, 3
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    o = dict(OPTS, maxiter=30)
    res = {}
    for ek in (5, 3):
        with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc=disc, eval_kernel=ek, tile_rows=(40 if ek == 5 else 0)) as pb:
            assert pb.info()["eval_kernel"] == ek
            res[ek] = pb.minimize_lbfgs(XP, 1.5 ** 6, o)
    for b in range(B):
        opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc=disc)
        x, A, st, nit, nfev = opb.minimize_lbfgs(XP[b], 1.5 ** 6, o)
        r = res[5]
        assert (r["nit"][b], r["nfev"][b], r["status"][b]) == (nit, nfev, st), b
        assert abs(r["A"][b] - A) <= 1e-6 * abs(A)
        assert np.abs(r["x"][b] - x).max() <= 1e-6
        assert (res[3]["nit"][b], res[3]["nfev"][b]) == (nit, nfev)


def test_short_ladder_matches_tile_kernel(capi):
    from varanneal_amd import twin
    D, N, B = 200, 200, 4
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    rf = 1.5 ** np.arange(4)
    out = {}
    for ek in (5, 3):
        with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", eval_kernel=ek, max_beta=4) as pb:
            out[ek] = pb.anneal(XP, rf, OPTS)
    assert np.allclose(out[5]["A"], out[3]["A"], rtol=1e-6, atol=0)
    assert np.allclose(out[5]["pest"], out[3]["pest"], rtol=1e-5, atol=0)


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_weight_arrays_and_sparse_data(capi, disc):
    """(N-1, D) model-error weights, (N_data, L) measurement weights, data at every nskip-th model time
    (va_ode.py:138-158, 197-222): the rows' weights travel through the ring with the rows; against the oracle"""
    import va_oracle
    from varanneal_amd import twin
    D, B = 200, 2
    for N, nskip, rm_arr, rf_arr, tile_rows in ((61, 1, True, True, 0), (121, 1, False, True, 40), (97, 2, False, False, 32),
                                                (91, 3, True, False, 30), (65, 2, True, True, 64)):
        rng = np.random.RandomState(N + nskip)
        L = 30
        Lidx = np.sort(rng.choice(D, L, replace=False))
        Nd = (N + nskip - 1) // nskip
        Y = rng.randn(Nd, L)
        RM = (0.5 + rng.rand(Nd, L)) if rm_arr else 3.0
        RF = (0.2 + rng.rand(N - 1, D)) if rf_arr else 0.7
        XP = np.concatenate([2.0 * rng.randn(B, N * D), 7.0 + rng.rand(B, 1)], axis=1)
        P = XP[:, -1:].copy()
        with capi.Problem(B, D, N, Y, list(Lidx), twin.DT, RM, RF, P, [0], disc=disc, eval_kernel=5, tile_rows=tile_rows,
                          merr_nskip=nskip) as pb:
            assert pb.info()["eval_kernel"] == 5, (N, nskip)
            A, me, fe, g = pb.action_grad(XP, 2.5)
        for b in range(B):
            opb = va_oracle.Problem(D, N, Y, list(Lidx), twin.DT, RM, RF, P[b], [0], disc=disc, merr_nskip=nskip)
            Ao, meo, feo, go = opb.action_grad(XP[b], 2.5)
            assert abs(A[b] - Ao) <= RTOL_A * abs(Ao), (disc, N, nskip, b, A[b], Ao)
            assert abs(me[b] - meo) <= RTOL_A * max(abs(meo), abs(Ao)) and abs(fe[b] - feo) <= RTOL_A * abs(feo), (disc, N, nskip, b)
            err = np.abs(g[b] - go)
            assert err.max() <= RTOL_G * np.abs(go).max(), (disc, N, nskip, b, int(err.argmax()), err.max())


def test_weight_arrays_minimisation_follows_the_tile_kernel(capi):
    """line-search launches with the weight images in the ring: same (nit, nfev) as the tile kernel, same end point"""
    from varanneal_amd import twin
    D, N, B = 200, 120, 3
    t, Y, _, Lidx = twin.make_twin(D, N)
    rng = np.random.RandomState(4)
    RF = 4e-6 * (0.5 + rng.rand(N - 1, D))
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    o = dict(OPTS, maxiter=25)
    res = {}
    for ek in (5, 3):
        with capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, RF, P, [0], disc="trapezoid", eval_kernel=ek, tile_rows=(40 if ek == 5 else 0)) as pb:
            assert pb.info()["eval_kernel"] == ek
            res[ek] = pb.minimize_lbfgs(XP, 1.5 ** 6, o)
    assert np.array_equal(res[5]["nit"], res[3]["nit"]) and np.array_equal(res[5]["nfev"], res[3]["nfev"])
    assert np.allclose(res[5]["A"], res[3]["A"], rtol=1e-9, atol=0) and np.abs(res[5]["x"] - res[3]["x"]).max() <= 1e-6


def test_random_shapes_against_the_oracle(capi):
    """a seeded sweep over widths, lengths, segment lengths, numbers of observed columns, weight kinds, data strides
    and discretisations -- every combination the streaming kernel accepts -- against the oracle"""
    import va_oracle
    from varanneal_amd import twin
    rng = np.random.RandomState(20260)
    discs = ["trapezoid", "SimpsonHermite", "euler", "forwardmap"]
    for case in range(36):
        D = 2 * int(rng.randint(33, 131))
        nskip = int(rng.choice([1, 1, 2, 3]))
        disc = discs[case % 4]
        Nd = int(rng.randint(18, 70))
        N = (Nd - 1) * nskip + 1
        if disc == "SimpsonHermite" and N % 2 == 0:
            Nd += 1; N = (Nd - 1) * nskip + 1
            if N % 2 == 0:                      # (nskip odd, Nd even -> N even again: take nskip = 2)
                nskip = 2; N = (Nd - 1) * nskip + 1
        B = int(rng.randint(1, 4))
        L = int(rng.randint(1, D // 2))
        Lidx = np.sort(rng.choice(D, L, replace=False))
        Y = rng.randn(Nd, L)
        RM = (0.5 + rng.rand(Nd, L)) if rng.rand() < 0.4 else 2.0
        RF = (0.2 + rng.rand(N - 1, D)) if rng.rand() < 0.4 else 0.9
        tile_rows = int(rng.choice([0, 32, 34, 48, N]))
        XP = np.concatenate([2.0 * rng.randn(B, N * D), 7.0 + rng.rand(B, 1)], axis=1)
        P = XP[:, -1:].copy()
        tag = (case, D, N, B, L, nskip, disc, tile_rows, np.ndim(RM), np.ndim(RF))
        with capi.Problem(B, D, N, Y, list(Lidx), twin.DT, RM, RF, P, [0], disc=disc, eval_kernel=5, tile_rows=tile_rows,
                          merr_nskip=nskip) as pb:
            assert pb.info()["eval_kernel"] == 5, tag
            A, me, fe, g = pb.action_grad(XP, 1.7)
        for b in range(B):
            opb = va_oracle.Problem(D, N, Y, list(Lidx), twin.DT, RM, RF, P[b], [0], disc=disc, merr_nskip=nskip)
            Ao, meo, feo, go = opb.action_grad(XP[b], 1.7)
            assert abs(A[b] - Ao) <= RTOL_A * abs(Ao), tag
            assert abs(me[b] - meo) <= RTOL_A * max(abs(meo), abs(Ao)) and abs(fe[b] - feo) <= RTOL_A * abs(feo), tag
            err = np.abs(g[b] - go)
            assert err.max() <= RTOL_G * np.abs(go).max(), tag + (int(err.argmax()), err.max())


@pytest.mark.parametrize("disc", ["SimpsonHermite", "euler"])
def test_line_search_launches_with_weights_and_sparse_data(capi, disc):
    """minimisation on the streaming kernel with everything at once: per-row RF and RM weights, data at every 2nd model
    time, an odd number of observed columns, several segments -- the iteration and evaluation counts of the tile kernel
    and of the oracle, the same end point"""
    import va_oracle
    from varanneal_amd import twin
    D, Nd, nskip, B = 200, 61, 2, 2
    N = (Nd - 1) * nskip + 1
    rng = np.random.RandomState(12)
    L = 41
    Lidx = np.sort(rng.choice(D, L, replace=False))
    t, Yfull, _, _ = twin.make_twin(D, N, Lidx=list(Lidx))
    Y = Yfull[::nskip]
    RM = 4.0 * (0.5 + rng.rand(Nd, L))
    RF = 4e-6 * (0.5 + rng.rand(N - 1, D))
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    o = dict(OPTS, maxiter=20)
    res = {}
    for ek in (5, 3):
        with capi.Problem(B, D, N, Y, list(Lidx), twin.DT / nskip, RM, RF, P, [0], disc=disc, eval_kernel=ek, merr_nskip=nskip,
                          tile_rows=(40 if ek == 5 else 0)) as pb:
            assert pb.info()["eval_kernel"] == ek
            res[ek] = pb.minimize_lbfgs(XP, 1.5 ** 4, o)
    assert np.array_equal(res[5]["nit"], res[3]["nit"]) and np.array_equal(res[5]["nfev"], res[3]["nfev"])
    for b in range(B):
        opb = va_oracle.Problem(D, N, Y, list(Lidx), twin.DT / nskip, RM, RF, P[b], [0], disc=disc, merr_nskip=nskip)
        x, A, st, nit, nfev = opb.minimize_lbfgs(XP[b], 1.5 ** 4, o)
        assert (res[5]["nit"][b], res[5]["nfev"][b], res[5]["status"][b]) == (nit, nfev, st), (disc, b)
        assert abs(res[5]["A"][b] - A) <= 1e-6 * abs(A) and np.abs(res[5]["x"][b] - x).max() <= 1e-6
