"""GPU: the drop-in Annealer end to end on the device (reference script flow:
examples/Lorenz96_D20/Lorenz96_anneal.py:75-92) against the golden ladders."""
import numpy as np
import pytest

from varanneal_amd import twin, va_ode

pytestmark = pytest.mark.gpu
OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


def _run(c, disc, fused, nb=None, **kw):
    a = va_ode.Annealer()
    a.set_model(twin.l96, int(c["D"]))
    a.set_data(c["Y"], t=c["t"])
    beta = c["beta"] if nb is None else c["beta"][:nb]
    a.anneal(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), beta, 4.0, 4e-6, list(c["Lidx"]), [0],
             dt_model=float(c["t"][1] - c["t"][0]), init_to_data=True, disc=disc, method='L-BFGS-B',
             opt_args=OPTS, adolcID=0, verbose=False, fused=fused, **kw)
    return a


@pytest.mark.parametrize("name,disc", [("g4_c1_trapezoid_N200", "trapezoid"),
                                        ("g4_shipped_SH_N161", "SimpsonHermite")])
def test_reference_script_flow(golden_ladders, name, disc, tmp_path):
    c = golden_ladders[name]
    a = _run(c, disc, fused=True)
    N, D, nb = int(c["N"]), int(c["D"]), len(c["beta"])
    assert np.all(np.abs(a.A_array[:12] - c["A_array"][:12]) <= 1e-8)
    assert abs(a.A_array[-1] - c["A_array"][-1]) <= 1e-3 * c["A_array"][-1]
    assert abs(a.P[0] - c["params"][-1, 0]) <= 2e-3 * abs(c["params"][-1, 0])
    assert list(a.nit_array[:7]) == list(c["nit"][:7])
    # the S1 evaluator agrees with the stored results at the final RF
    A, g = a.A_gradA_taped(a.minpaths[-1])
    assert abs(A - a.A_array[-1]) <= 1e-12 * A and np.abs(g).max() < 1e-3
    a.save_paths(str(tmp_path / "p.npy"))
    assert np.load(str(tmp_path / "p.npy")).shape == (nb, N, D + 1)
    a.close()


def test_stepwise_equals_fused(golden_ladders):
    c = golden_ladders["g4_c1_trapezoid_N200"]
    f = _run(c, "trapezoid", fused=True, nb=14)
    s = _run(c, "trapezoid", fused=False, nb=14)
    assert np.array_equal(f.A_array, s.A_array) and np.array_equal(f.minpaths, s.minpaths)
    assert np.array_equal(f.nfev_array, s.nfev_array)
    f.close(); s.close()


def test_bounds_route_uses_device_evaluator(golden_ladders):
    c = golden_ladders["g4_c1_trapezoid_N200"]
    bounds = [(-15.0, 15.0)] * 20 + [(6.5, 10.0)]
    a = _run(c, "trapezoid", fused=None, nb=16, bounds=bounds)
    assert np.all(a.minpaths[:, -1] >= 6.5 - 1e-12)
    assert np.all(np.abs(a.minpaths[:, :-1]) <= 15.0 + 1e-12)
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12) and np.all(a.exitflags == 0)
    A, g = a.A_gradA_taped(a.minpaths[-1])                            # SciPy's f is the device's f
    assert abs(A - a.A_array[-1]) <= 1e-12 * A
    a.close()


def test_batched_seeds_independent():
    D, N, B, nb = 20, 200, 6, 10
    t, Y, _, Lidx = twin.make_twin(D, N)
    X0 = np.empty((B, N, D)); P0 = np.empty((B, 1))
    for b in range(B):
        X0[b], P0[b] = twin.initial_guess(N, D, b)
    a = va_ode.Annealer(); a.set_model("lorenz96", D); a.set_data(Y, t=t)
    a.anneal(X0.copy(), P0.copy(), 1.5, np.arange(nb), 4.0, 4e-6, Lidx, [0], opt_args=OPTS, verbose=False)
    s = va_ode.Annealer(); s.set_model("lorenz96", D); s.set_data(Y, t=t)
    s.anneal(X0[4].copy(), P0[4].copy(), 1.5, np.arange(nb), 4.0, 4e-6, Lidx, [0], opt_args=OPTS, verbose=False)
    assert np.array_equal(a.A_array[4], s.A_array) and np.array_equal(a.minpaths[4], s.minpaths)
    a.close(); s.close()


def test_c_abi_gather_of_result_tables():
    """va_gather_results (the RCCL all-gather of SURVEY.md 8(b)/(e)) on a one-rank communicator: the
    gathered table is the table va_anneal returned.  (More ranks need more GPUs than this box has;
    the multi-rank layout is covered on CPU by tests/test_parallel_gloo.py.)"""
    from varanneal_amd import _capi, twin
    D, N, B, nb = 20, 120, 5, 4
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", max_beta=nb) as pb:
        r = pb.anneal(XP, 1.5 ** np.arange(nb), {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 100000, 'maxiter': 20})
        comm = _capi.Comm(_capi.Comm.unique_id(), 1, 0, device=0)
        table, st = pb.gather_results(comm, nb)
        comm.close()
    assert table.shape == (B, nb, 4) and np.array_equal(table[:, :, 0], r["A"]) and np.array_equal(table[:, :, 1], r["me"])
    assert np.array_equal(table[:, :, 2], r["fe"]) and np.array_equal(table[:, :, 3], r["pest"][:, :, 0])
    assert np.array_equal(st, r["status"])


def test_wide_models_run_the_streaming_kernel_through_the_drop_in():
    """set_model(l96, D = 100): the library's own streaming kernel (k_eval5, run-time width; no module is generated
    for a built-in model).  A user's stencil that is NOT in the registry -- Lorenz-96 with a damping parameter -- is
    traced, gets its column form compiled for k_eval5, and with the damping fixed at 1 one rung agrees with the
    built-in's to 1e-9 (same arithmetic up to one multiplication by 1 and the order of a few additions)."""
    from varanneal_amd import twin, va_ode
    D, N, B = 100, 160, 4
    Lidx = list(range(0, D, 5))
    t, Y, _, _ = twin.make_twin(D, N, Lidx=Lidx)
    X0 = np.empty((B, N, D)); P0 = np.empty((B, 1))
    for s in range(B):
        X0[s], P0[s] = twin.initial_guess(N, D, s, Y, Lidx)

    def damped_l96(t, x, p):
        return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[1] * x + p[0]
    res = {}
    for key, f, P in (("builtin", twin.l96, P0), ("traced", damped_l96, np.hstack([P0, np.ones((B, 1))]))):
        a = va_ode.Annealer()
        a.set_model(f, D)
        a.set_data(Y, t=t)
        a.anneal(X0.copy(), P.copy(), 1.5, [0], 4.0, 4e-6, Lidx, [0], disc="trapezoid",
                 opt_args={'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': 20, 'maxfun': 1000}, verbose=False)
        assert a._pb.info()["eval_kernel"] == 5
        mod = getattr(a, "_rhs_module", None)
        assert (mod is not None) == (key == "traced")
        if mod is not None:
            assert mod["col_variant"][0] == 5
        res[key] = (a.A_array.copy(), a.nit_array.copy())
        a.close()
    assert np.array_equal(res["builtin"][1], res["traced"][1]) and np.allclose(res["builtin"][0], res["traced"][0], rtol=1e-9)


def test_wide_models_with_per_component_weights_and_simpson_hermite():
    """the same two models through the drop-in with what the tutorials use on top: RF0 of shape (D,) (one model-error
    weight per component, va_ode.py:631-634), RM of shape (L,), Simpson-Hermite, an ODD number of observed columns
    -- all on the streaming kernel (weights through the ring, interval walk, pad column); built-in and traced
    agree, and the first rung agrees with the oracle's minimisation"""
    import va_oracle
    from varanneal_amd import twin, va_ode
    D, N, B = 100, 161, 3
    Lidx = list(range(0, D, 5))[:19]
    t, Y, _, _ = twin.make_twin(D, N, Lidx=Lidx)
    X0 = np.empty((B, N, D)); P0 = np.empty((B, 1))
    for s in range(B):
        X0[s], P0[s] = twin.initial_guess(N, D, s, Y, Lidx)
    RF0 = 4e-6 * (1.0 + 0.05 * np.arange(D))
    RM = 4.0 * (1.0 + 0.1 * np.arange(len(Lidx)))
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': 15, 'maxfun': 1000}

    def damped_l96(t, x, p):
        return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[1] * x + p[0]
    res = {}
    for key, f, P in (("builtin", twin.l96, P0), ("traced", damped_l96, np.hstack([P0, np.ones((B, 1))]))):
        a = va_ode.Annealer()
        a.set_model(f, D)
        a.set_data(Y, t=t)
        a.anneal(X0.copy(), P.copy(), 1.5, [0, 1], RM, RF0, Lidx, [0], disc="SimpsonHermite", opt_args=opts, verbose=False)
        assert a._pb.info()["eval_kernel"] == 5
        res[key] = (a.A_array.copy(), a.nit_array.copy(), a.minpaths.copy() if hasattr(a, "minpaths") else None)
        a.close()
    assert np.array_equal(res["builtin"][1], res["traced"][1]) and np.allclose(res["builtin"][0], res["traced"][0], rtol=1e-9)
    XP = np.append(X0[0].ravel(), P0[0])
    XP[:N * D].reshape(N, D)[:, Lidx] = Y                                   # init_to_data (va_ode.py:677-678)
    opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, np.resize(RM, Y.shape), np.resize(RF0, (N - 1, D)), P0[0], [0],
                            disc="SimpsonHermite")
    x, A, st, nit, nfev = opb.minimize_lbfgs(XP, 1.0, opts)
    assert nit == res["builtin"][1][0][0] and abs(A - res["builtin"][0][0][0]) <= 1e-7 * abs(A)


def test_seed_sharding_two_ranks_on_one_card(tmp_path):
    """examples/Lorenz96_D20/Lorenz96_multi_gpu.py as a 2-rank job (`--share-gpu`: both ranks on cuda:0, the closing
    gather over gloo) against the same seeds in one process: every seed's ladder is identical to the last bit whichever
    rank, and whichever batch, annealed it (the reference's array job: submit_multiM.sh:14-30)"""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "examples", "Lorenz96_D20", "Lorenz96_multi_gpu.py")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--seeds", "5", "--N", "100", "--nbeta", "4"]
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    # (both runs with --share-gpu: a process group over gloo -- RCCL's bootstrap alone can take minutes on a box without a network)
    r = subprocess.run([sys.executable, script] + args + ["--share-gpu", "--out", one], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script] + args + ["--share-gpu", "--out", two], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    for k in ("A", "flags", "k", "nfev"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    assert a["A"].shape == (5, 4)


def test_wide_model_with_a_finer_model_grid_through_the_drop_in():
    """dt_model = dt_data / 2 at D = 100 (va_ode.py:555-558: a measurement at every second model time), trapezoid and
    Simpson-Hermite: the streaming kernel (data rows by quotient / remainder, weight 0 between measurements); first
    rungs against the oracle's minimisation"""
    import va_oracle
    from varanneal_amd import twin, va_ode
    D, Nd, B = 100, 61, 2
    N = 2 * (Nd - 1) + 1
    Lidx = list(range(0, D, 4))
    t, Yfull, _, _ = twin.make_twin(D, N, Lidx=Lidx, dt=twin.DT / 2)
    Y, td = Yfull[::2], t[::2]
    X0 = np.empty((B, N, D)); P0 = np.empty((B, 1))
    for s in range(B):
        X0[s], P0[s] = twin.initial_guess(N, D, s, Y, Lidx, nskip=2)
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': 12, 'maxfun': 1000}
    for disc in ("trapezoid", "SimpsonHermite"):
        a = va_ode.Annealer()
        a.set_model(twin.l96, D)
        a.set_data(Y, t=td)
        a.anneal(X0.copy(), P0.copy(), 1.5, [0, 1], 4.0, 4e-6, Lidx, [0], dt_model=twin.DT / 2, disc=disc, opt_args=opts, verbose=False)
        assert a._pb.info()["eval_kernel"] == 5 and a.merr_nskip == 2 and a.N_model == N
        A_dev, nit_dev = a.A_array.copy(), a.nit_array.copy()
        a.close()
        for s in range(B):
            opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT / 2, 4.0, 4e-6, P0[s], [0], disc=disc, merr_nskip=2)
            x, A, st, nit, nfev = opb.minimize_lbfgs(np.append(X0[s].ravel(), P0[s]), 1.0, opts)
            assert nit == nit_dev[s][0] and abs(A - A_dev[s][0]) <= 1e-7 * abs(A), (disc, s)
