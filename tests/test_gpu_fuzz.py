"""GPU: seeded random problem shapes through the C-ABI against the C oracle -- state sizes on
both sides of every kernel-geometry switch (column-run 256/512/1024-thread groups, flat kernel),
all discretisations, dt_model = dt_data / nskip, scalar and array RM / RF0, random observed
subsets, 1-5 seeds; every case also takes a few device L-BFGS iterations step for step."""
import numpy as np
import pytest

import va_oracle
from varanneal_amd import _capi

pytestmark = pytest.mark.gpu
OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 4}


def _case(seed):
    rng = np.random.RandomState(7000 + seed)
    D = int(rng.choice([4, 5, 6, 9, 16, 20, 21, 33, 64, 65, 100, 128, 200, 255, 256, 257, 300, 511, 513, 700, 1200]))
    disc = str(rng.choice(["euler", "trapezoid", "SimpsonHermite", "forwardmap"]))
    nskip = int(rng.choice([1, 1, 2, 3]))
    N_data = int(rng.randint(2, 90 if D > 128 else 220))
    N = (N_data - 1) * nskip + 1
    if disc == "SimpsonHermite" and N % 2 == 0:
        N_data += 1
        N = (N_data - 1) * nskip + 1
        if N % 2 == 0:                       # nskip even keeps N odd; otherwise fall back
            disc = "trapezoid"
    L = int(rng.randint(1, min(D, 12) + 1))
    Lidx = sorted(rng.choice(D, L, replace=False).tolist())
    B = int(rng.randint(1, 6))
    Y = rng.randn(N_data, L)
    RM = 0.5 + rng.rand(N_data, L) if rng.rand() < 0.4 else float(1.0 + rng.rand())
    RF0 = 0.2 + rng.rand(N - 1, D) if rng.rand() < 0.4 else float(0.1 + rng.rand())
    XP = np.concatenate([2.0 * rng.randn(B, N * D), 6.0 + 3.0 * rng.rand(B, 1)], axis=1)
    est = rng.rand() < 0.8
    return dict(D=D, N=N, disc=disc, nskip=nskip, Lidx=Lidx, B=B, Y=Y, RM=RM, RF0=RF0, XP=XP, est=est,
                rf=float(10.0 ** rng.uniform(-2, 2)))


@pytest.mark.parametrize("seed", range(120))
def test_random_problem(seed):
    c = _case(seed)
    D, N, B = c["D"], c["N"], c["B"]
    Pidx = [0] if c["est"] else []
    P = c["XP"][:, -1:].copy()
    XP = c["XP"] if c["est"] else c["XP"][:, :-1].copy()
    pb = _capi.Problem(B, D, N, c["Y"], c["Lidx"], 0.025, c["RM"], c["RF0"], P, Pidx, disc=c["disc"],
                       merr_nskip=c["nskip"])
    A, me, fe, g = pb.action_grad(XP, c["rf"])
    r = pb.minimize_lbfgs(XP, c["rf"], OPTS)
    info = pb.info()
    pb.close()
    for b in range(B):
        opb = va_oracle.Problem(D, N, c["Y"], c["Lidx"], 0.025, c["RM"], c["RF0"], P[b], Pidx, disc=c["disc"],
                                merr_nskip=c["nskip"])
        Ao, meo, feo, go = opb.action_grad(XP[b], c["rf"])
        tag = (seed, D, N, c["disc"], c["nskip"], info)
        assert abs(A[b] - Ao) <= 1e-12 * abs(Ao) and abs(me[b] - meo) <= 1e-12 * abs(Ao), tag
        assert np.abs(g[b] - go).max() <= 1e-10 * np.abs(go).max(), tag
        x, Am, st, nit, nfev = opb.minimize_lbfgs(XP[b], c["rf"], OPTS)
        assert (r["nit"][b], r["nfev"][b], r["status"][b]) == (nit, nfev, st), tag
        assert abs(r["A"][b] - Am) <= 1e-8 * abs(Am), tag


def test_too_wide_state_is_refused_cleanly():
    """the flat kernel stages (T + halo) rows of 3 arrays in LDS: beyond ~1600 columns not even two
    owned rows fit the CU's 160 KiB and the problem is refused with a message, not a fault"""
    D, N = 4000, 5
    rng = np.random.RandomState(0)
    with pytest.raises(_capi.VaError) as e:
        _capi.Problem(1, D, N, rng.randn(N, 2), [0, 7], 0.025, 1.0, 1.0, np.ones((1, 1)), [0])
    assert e.value.code == -4 and "too wide" in str(e.value)


@pytest.mark.parametrize("seed", range(60))
def test_random_network(seed):
    """random layer structures (1-80 neurons, 2-6 layers), 1-90 examples, every activation, random
    estimated-parameter subsets and observed neurons, scalar or [in, out] RM, 1-3 seeds: both
    evaluators (single-kernel and tiled) against the NumPy oracle"""
    import va_nnet_oracle as vno
    rng = np.random.RandomState(9000 + seed)
    nl = int(rng.randint(2, 7))
    wmax = int(rng.choice([6, 20, 32, 33, 80]))
    structure = [int(rng.randint(1, wmax + 1)) for _ in range(nl)]
    M = int(rng.choice([1, 2, 7, 32, 33, 90]))
    act = str(rng.choice(["sigmoid", "tanh", "linear", "relu", "softplus"]))
    B = int(rng.randint(1, 4))
    Lin = sorted(rng.choice(structure[0], rng.randint(1, structure[0] + 1), replace=False).tolist())
    Lout = sorted(rng.choice(structure[-1], rng.randint(1, structure[-1] + 1), replace=False).tolist())
    din, dout = rng.randn(M, len(Lin)), rng.rand(M, len(Lout))
    NP = sum(structure[n + 1] * structure[n] + structure[n + 1] for n in range(nl - 1))
    npest = int(rng.randint(0, NP + 1))
    Pidx = sorted(rng.choice(NP, npest, replace=False).tolist())
    RM = [1.0 + rng.rand(), 2.0 + rng.rand()] if rng.rand() < 0.5 else float(1.0 + rng.rand())
    P = 0.4 * rng.randn(B, NP)
    X = rng.rand(B, M * sum(structure))
    XP = np.concatenate([X, P[:, Pidx]], axis=1)
    pr = _capi.NnetProblem(B, structure, din, dout, [Lin, Lout], RM, 0.3, P, Pidx, act=act)
    A, me, fe, g = pr.action_grad(XP, 2.0)
    pr.close()
    for b in range(B):
        pb = vno.NnetProblem(structure, din, dout, [Lin, Lout], np.asarray(RM) if isinstance(RM, list) else RM, 0.3,
                             P[b], Pidx, act=act)
        A1, me1, fe1, g1 = pb.action_grad(XP[b], 2.0)
        tag = (seed, structure, M, act, npest)
        assert abs(A[b] - A1) <= 1e-12 * abs(A1) and abs(me[b] - me1) <= 1e-12 * abs(A1), tag
        assert np.abs(g[b] - g1).max() <= 1e-10 * max(np.abs(g1).max(), 1e-300), tag


@pytest.mark.parametrize("seed", range(40))
def test_random_flat_kernel_features(seed):
    """time-dependent parameters and/or full RM matrices (both flat-kernel features) on random
    shapes: value against the NumPy restatement, gradient through complex-step directional
    derivatives of that restatement along three random directions"""
    from varanneal_amd import twin
    rng = np.random.RandomState(11000 + seed)
    D = int(rng.choice([4, 5, 12, 20, 37, 70]))
    tdp = rng.rand() < 0.7
    rmfull = (not tdp) or rng.rand() < 0.5
    disc = str(rng.choice(["trapezoid", "SimpsonHermite"] if tdp else ["euler", "trapezoid", "SimpsonHermite", "forwardmap"]))
    nskip = int(rng.choice([1, 1, 2]))
    N_data = int(rng.randint(2, 70))
    N = (N_data - 1) * nskip + 1
    if disc == "SimpsonHermite" and N % 2 == 0:
        disc = "trapezoid"
    L = int(rng.randint(1, min(D, 6) + 1))
    Lidx = sorted(rng.choice(D, L, replace=False).tolist())
    Y = rng.randn(N_data, L)
    if rmfull:
        RM = np.array([2.0 * np.eye(L) + 0.5 * rng.randn(L, L) for _ in range(N_data)])
    else:
        RM = 0.5 + rng.rand(N_data, L) if rng.rand() < 0.5 else float(1.0 + rng.rand())
    RF0 = 0.2 + rng.rand(N - 1, D) if rng.rand() < 0.5 else float(0.1 + rng.rand())
    P = 6.0 + 3.0 * rng.rand(N, 1) if tdp else 6.0 + 3.0 * rng.rand(1)
    X = 2.0 * rng.randn(N * D)
    XP = np.append(X, P.ravel())
    rf = float(10.0 ** rng.uniform(-1, 1))
    RFs = RF0 * rf
    fun = lambda z: va_oracle.numpy_action_generic(twin.l96, z, D, N, Y, Lidx, 0.025, RM, RFs, 1, [0], P, disc,
                                                   nskip=nskip)
    A0, me0, fe0 = fun(XP)
    pb = _capi.Problem(1, D, N, Y, Lidx, 0.025, RM, RF0, P[None] if tdp else P[None, :], [0], disc=disc,
                       merr_nskip=nskip, p_time_dependent=tdp)
    A, me, fe, g = pb.action_grad(XP[None, :], rf)
    pb.close()
    tag = (seed, D, N, disc, nskip, tdp, rmfull)
    assert abs(A[0] - A0) <= 1e-12 * abs(A0) and abs(me[0] - me0) <= 1e-12 * abs(A0), tag
    for _ in range(3):
        v = rng.randn(XP.size)
        dd = np.imag(fun(XP + 1e-30j * v)[0]) / 1e-30
        assert abs(np.dot(g[0], v) - dd) <= 1e-10 * (np.abs(g[0]) * np.abs(v)).sum(), tag


@pytest.mark.parametrize("seed", range(24))
def test_random_ladder(seed):
    """whole RF ladders (6 rungs) on random small problems -- the per-seed ladder bookkeeping on the
    device, with seeds at different rungs at the same time.  Rugged random data make rung-level
    agreement with ANY other implementation a matter of luck beyond the first iterations (the
    golden twin ladders cover that); what must hold exactly: a seed annealed alone gives bit for bit
    what it gives inside a batch, every stored minimiser re-evaluates to the stored action, and the
    first rung starts out on the oracle's iterates."""
    rng = np.random.RandomState(13000 + seed)
    D = int(rng.choice([5, 8, 20, 33]))
    disc = str(rng.choice(["euler", "trapezoid", "SimpsonHermite", "forwardmap"]))
    N = int(rng.randint(6, 60))
    if disc == "SimpsonHermite" and N % 2 == 0:
        N += 1
    L = int(rng.randint(1, min(D, 6) + 1))
    Lidx = sorted(rng.choice(D, L, replace=False).tolist())
    B = int(rng.randint(2, 6))
    Y = 2.0 * rng.randn(N, L)
    XP = np.concatenate([2.0 * rng.randn(B, N * D), 6.0 + 3.0 * rng.rand(B, 1)], axis=1)
    XP[:, :N * D].reshape(B, N, D)[:, :, Lidx] = Y
    P = XP[:, -1:].copy()
    nb = 6
    rfs = 2.0 ** np.arange(nb)
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 100000}
    pb = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 4e-4, P, [0], disc=disc, max_beta=nb, keep_paths=1)
    r = pb.anneal(XP, rfs, opts, want_paths=True)
    tag = (seed, D, N, disc, B)
    assert np.all(r["status"] == 0), tag
    for k in (0, nb - 1):                                   # stored minimisers re-evaluate to the stored actions
        xk = np.concatenate([r["minpaths"][:, k, :N * D], r["pest"][:, k, :]], axis=1)
        A, me, fe, _ = pb.action_grad(xk, rfs[k], want_grad=False)
        assert np.all(np.abs(A - r["A"][:, k]) <= 1e-12 * A) and np.all(np.abs(me - r["me"][:, k]) <= 1e-12 * A), tag
    pb.close()
    b = int(rng.randint(B))                                 # one seed alone == the same seed in the batch
    p1 = _capi.Problem(1, D, N, Y, Lidx, 0.025, 4.0, 4e-4, P[b:b + 1], [0], disc=disc, max_beta=nb, keep_paths=1)
    r1 = p1.anneal(XP[b:b + 1], rfs, opts, want_paths=True)
    p1.close()
    assert np.array_equal(r1["A"][0], r["A"][b]) and np.array_equal(r1["minpaths"][0], r["minpaths"][b]), tag
    assert np.array_equal(r1["nfev"][0], r["nfev"][b]), tag
    # the first ten iterations of the first rung against the oracle's restated L-BFGS-B
    o10 = dict(opts, maxiter=10)
    p2 = _capi.Problem(1, D, N, Y, Lidx, 0.025, 4.0, 4e-4, P[b:b + 1], [0], disc=disc)
    r2 = p2.minimize_lbfgs(XP[b:b + 1], 1.0, o10)
    p2.close()
    opb = va_oracle.Problem(D, N, Y, Lidx, 0.025, 4.0, 4e-4, P[b], [0], disc=disc)
    x, Am, st, nit, nfev = opb.minimize_lbfgs(XP[b], 1.0, o10)
    assert (r2["nit"][0], r2["nfev"][0], r2["status"][0]) == (nit, nfev, st), tag
    assert abs(r2["A"][0] - Am) <= 1e-9 * abs(Am), tag


def _random_model(seed):
    """a smooth random right-hand side mixing the functions user models use (products, powers,
    exp / tanh / sqrt / division), with NP parameters and optionally a stimulus"""
    rng = np.random.RandomState(15000 + seed)
    D = int(rng.randint(2, 7)); NP = int(rng.randint(1, 6)); stim = bool(rng.rand() < 0.5)
    A = rng.randn(D, D) * 0.3
    c = rng.randn(D) * 0.5
    kinds = rng.randint(0, 5, size=D)
    pk = rng.randint(0, NP, size=(D, 2))

    def f(t, x, pin):
        p, s = pin if stim else (pin, None)
        out = np.zeros_like(x)
        for i in range(D):
            lin = sum(A[i, j] * x[:, j] for j in range(D))
            pa, pb = p[pk[i, 0]], p[pk[i, 1]]
            xi, xn = x[:, i], x[:, (i + 1) % D]
            if kinds[i] == 0:
                term = pa * xi * xn + c[i] * xi ** 3
            elif kinds[i] == 1:
                term = pa * np.tanh(xn / (1.0 + pb ** 2)) - xi
            elif kinds[i] == 2:
                term = np.exp(-0.1 * xi ** 2) * pa + pb * xn
            elif kinds[i] == 3:
                term = pa / (1.0 + xn ** 2) + np.sqrt(1.0 + xi ** 2) * pb
            else:
                term = pa * np.sin(0.3 * t) * xn - pb * xi
            out[:, i] = lin + term + (s * c[i] if stim else 0.0)
        return out
    return f, D, NP, stim


@pytest.mark.parametrize("seed", range(6))
def test_random_generated_model(seed):
    """codegen on random models (trace -> SymPy derivatives -> HIP module built on the spot):
    value against the NumPy restatement with the ORIGINAL callable, gradient through complex-step
    directional derivatives; static and time-dependent parameters"""
    from varanneal_amd import codegen
    f, D, NP, stim = _random_model(seed)
    rng = np.random.RandomState(16000 + seed)
    tdp = seed % 2 == 1
    if tdp:                                       # time-dependent form: p[:, k]
        g0 = f
        f = lambda t, x, pin: g0(t, x, (tuple(pin[0].T), pin[1]) if stim else tuple(pin.T))
    N = int(rng.choice([21, 41]))
    disc = str(rng.choice(["trapezoid", "SimpsonHermite"]))
    t = 0.05 * np.arange(N)
    st = rng.randn(N) if stim else None
    mod = codegen.module_for(f, D, NP, 1 if stim else 0, 1, p_rows=tdp)
    rid = _capi.load_rhs_module(mod["so"])
    L = int(rng.randint(1, D + 1))
    Lidx = sorted(rng.choice(D, L, replace=False).tolist())
    Y = rng.randn(N, L)
    npe = int(rng.randint(1, NP + 1))
    Pidx = sorted(rng.choice(NP, npe, replace=False).tolist())
    P = 0.5 + rng.rand(N, NP) if tdp else 0.5 + rng.rand(NP)
    X = 0.8 * rng.randn(N * D)
    XP = np.append(X, (P[:, Pidx] if tdp else P[Pidx]).ravel())
    fun = lambda z: va_oracle.numpy_action_generic(f, z, D, N, Y, Lidx, 0.05, 2.0, 0.7, NP, Pidx, P, disc, t_model=t,
                                                   stim=st)
    A0, me0, fe0 = fun(XP)
    pb = _capi.Problem(1, D, N, Y, Lidx, 0.05, 2.0, 0.7, P[None] if tdp else P[None, :], Pidx, disc=disc, rhs=rid,
                       t_model=t, stim=st, p_time_dependent=tdp)
    A, me, fe, g = pb.action_grad(XP[None, :], 1.0)
    pb.close()
    tag = (seed, D, NP, stim, tdp, disc)
    assert abs(A[0] - A0) <= 1e-12 * abs(A0), tag
    for _ in range(3):
        v = rng.randn(XP.size)
        dd = np.imag(fun(XP + 1e-30j * v)[0]) / 1e-30
        assert abs(np.dot(g[0], v) - dd) <= 1e-10 * (np.abs(g[0]) * np.abs(v)).sum(), tag
