"""CPU: the __host__ __device__ core of the HIP path (varanneal_amd/csrc/va_core.h --
tile phases, halo/edge indexing, More'-Thuente step, coefficient-space two-loop,
per-seed ladder state machine) driven serially by tests/cpu_emul and checked against
the golden vectors and the oracle.  The emulator is test infrastructure only."""
import numpy as np
import pytest

import va_oracle
from _util import rm_rf_for
from cpu_emul import emul
from varanneal_amd import _capi

OPTS = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}


def _desc(c, batch=1, eval_kernel=0):
    N, D = int(c["N_model"]), int(c["D"])
    RM, RF0 = rm_rf_for(c)
    return _capi.make_desc(batch, D, N, c["Y"], c["Lidx"], c["dt_model"], RM, RF0,
                           np.tile(c["XP"][N * D:], (batch, 1)), [0], disc=str(c["disc"]),
                           merr_nskip=int(c["merr_nskip"]), eval_kernel=eval_kernel)


@pytest.mark.parametrize("sub", [2, 3])
def test_wave_private_tiles_with_several_subtiles_per_wave(golden_single, sub, monkeypatch):
    """k_eval4 with SUB sub-tiles per wave (one wave per SIMD on small grids): same results."""
    monkeypatch.setenv("VA_EMUL_SUB", str(sub))
    for name, c in golden_single.items():
        desc, keep = _desc(c, eval_kernel=4)
        for T in (72, 84):
            A, me, fe, g = emul.action_grad(desc, T, c["XP"][None, :], c["rf_scale"])
            assert abs(A[0] - c["A"]) <= 1e-12 * abs(c["A"]), (name, T)
            if "grad" in c:
                assert np.abs(g[0] - c["grad"]).max() <= 1e-11 * np.abs(c["grad"]).max(), (name, T)


@pytest.mark.parametrize("eval_kernel", [1, 3, 4])
def test_tile_phases_match_golden_for_every_tiling(golden_single, eval_kernel):
    """flat-mapped (va_core.h), column-run (va_tile3.h) and wave-private column-run (va_tile4.h)
    tiles, every discretisation, scalar/vector RM/RF, nskip 1 and 2."""
    for name, c in golden_single.items():
        desc, keep = _desc(c, eval_kernel=eval_kernel)
        for T in (2, 7, 50, 72, 84, 144, 400):   # tiny tiles, ragged last tile, K = 5, 6, 7, 12 runs, single tile
            A, me, fe, g = emul.action_grad(desc, T, c["XP"][None, :], c["rf_scale"])
            assert abs(A[0] - c["A"]) <= 1e-12 * abs(c["A"]), (name, T)
            assert abs(me[0] - c["me"]) <= 1e-12 * max(abs(c["me"]), abs(c["A"])), (name, T)
            if "grad" in c:
                assert np.abs(g[0] - c["grad"]).max() <= 1e-11 * np.abs(c["grad"]).max(), (name, T)


def _c1(golden_ladders, name):
    c = golden_ladders[name]
    N, D = int(c["N"]), int(c["D"])
    X0 = c["X0"].copy(); X0[:, c["Lidx"]] = c["Y"]
    XP0 = np.append(X0.flatten(), c["P0"])
    desc, keep = _capi.make_desc(1, D, N, c["Y"], c["Lidx"], 0.025, 4.0, 4e-6, c["P0"][None, :], [0],
                                 disc=str(c["disc"]))
    opb = va_oracle.Problem(D, N, c["Y"], c["Lidx"], 0.025, 4.0, 4e-6, c["P0"], [0], disc=str(c["disc"]))
    return c, N, D, XP0, desc, keep, opb


@pytest.mark.parametrize("name", ["g4_c1_trapezoid_N200", "g4_shipped_SH_N161"])
def test_state_machine_matches_oracle_step_for_step(golden_ladders, name):
    c, N, D, XP0, desc, keep, opb = _c1(golden_ladders, name)
    for rf, k in ((1.0, 3), (1.5 ** 7, 6), (1.5 ** 7, 10 ** 6), (1.5 ** 14, 20)):
        o = dict(OPTS, maxiter=k)
        r = emul.anneal(desc, 50, XP0[None, :], [rf], o)
        x, A, st, nit, nfev = opb.minimize_lbfgs(XP0, rf, o)
        assert (r["nit"][0, 0], r["nfev"][0, 0], r["status"][0, 0]) == (nit, nfev, st), (rf, k)
        assert abs(r["A"][0, 0] - A) <= 1e-6 * abs(A)
        assert np.abs(r["x"][0] - x).max() <= 1e-6


@pytest.mark.parametrize("name", ["g4_c1_trapezoid_N200", "g4_shipped_SH_N161"])
def test_ladder_lands_in_reference_basin(golden_ladders, name):
    c, N, D, XP0, desc, keep, opb = _c1(golden_ladders, name)
    rf = 1.5 ** c["beta"].astype(np.uint16)
    r = emul.anneal(desc, 50, XP0[None, :], rf, OPTS)
    assert np.all(np.abs(r["A"][0, :12] - c["A_array"][:12]) <= 1e-8)
    assert list(r["nit"][0, :7]) == list(c["nit"][:7])
    assert abs(r["A"][0, -1] - c["A_array"][-1]) <= 1e-3 * c["A_array"][-1]
    assert abs(r["pest"][0, -1, 0] - c["params"][-1, 0]) <= 2e-3 * abs(c["params"][-1, 0])
    assert np.array_equal(r["minpaths"][0, -1, :N * D], r["x"][0, :N * D])
    assert np.array_equal(r["minpaths"][0, :, N * D], r["pest"][0, :, 0])


def test_small_history_and_limits(golden_ladders):
    c, N, D, XP0, desc, keep, opb = _c1(golden_ladders, "g4_c1_trapezoid_N200")
    # maxcor=3: the circular history wraps many times; still tracks the oracle
    o = dict(OPTS, maxcor=3, maxiter=40)
    r = emul.anneal(desc, 50, XP0[None, :], [1.5 ** 12], o)
    x, A, st, nit, nfev = opb.minimize_lbfgs(XP0, 1.5 ** 12, o)
    assert (r["nit"][0, 0], r["nfev"][0, 0], r["status"][0, 0]) == (nit, nfev, st)
    assert abs(r["A"][0, 0] - A) <= 1e-6 * abs(A)
    # gtol satisfied at the start: zero iterations, one evaluation
    r = emul.anneal(desc, 50, XP0[None, :], [1.0], dict(OPTS, gtol=1e3))
    assert (r["nit"][0, 0], r["nfev"][0, 0], r["status"][0, 0]) == (0, 1, 0)
    assert np.array_equal(r["x"][0], XP0)


@pytest.mark.parametrize("D,disc", [(7, "trapezoid"), (36, "SimpsonHermite"), (64, "euler"),
                                    (100, "trapezoid"), (200, "SimpsonHermite"), (131, "SimpsonHermite"),
                                    (256, "forwardmap")])
def test_other_state_sizes_against_oracle(D, disc):
    """odd D (scalar staging), 256-thread groups up to D=64, 1024-thread groups beyond."""
    from varanneal_amd import twin
    N = 31
    t, Y, _, Lidx = twin.make_twin(D, N)
    rng = np.random.RandomState(D)
    XP = np.append(rng.randn(N * D) * 3.0, 7.3)
    RF0 = 4e-6 * (0.5 + rng.rand(N - 1, D))
    opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, RF0, [7.3], [0], disc=disc)
    Ao, meo, feo, go = opb.action_grad(XP, 1.5 ** 20)
    for ek in (1, 3, 4):
        desc, keep = _capi.make_desc(1, D, N, Y, Lidx, twin.DT, 4.0, RF0, [[7.3]], [0], disc=disc, eval_kernel=ek)
        for T in (4, 10):
            A, me, fe, g = emul.action_grad(desc, T, XP[None, :], 1.5 ** 20)
            assert abs(A[0] - Ao) <= 1e-12 * abs(Ao), (ek, T)
            assert np.abs(g[0] - go).max() <= 1e-11 * np.abs(go).max(), (ek, T)


@pytest.mark.parametrize("eval_kernel", [1, 3, 4])
def test_lidx_in_any_order(eval_kernel):
    """data column l pairs with state column Lidx[l] whatever the order of Lidx (va_ode.py:141)."""
    rng = np.random.RandomState(5)
    for D, Lidx in ((8, [5, 1, 3]), (20, [16, 0, 8, 2, 14, 4, 10])):
        N = 50
        Y = rng.randn(N, len(Lidx))
        XP = np.append(2.0 * rng.randn(N * D), 7.3)
        opb = va_oracle.Problem(D, N, Y, Lidx, 0.025, 3.0, 0.7, [7.3], [0], disc="trapezoid")
        Ao, meo, feo, go = opb.action_grad(XP, 2.5)
        desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 3.0, 0.7, [[7.3]], [0], disc="trapezoid",
                                     eval_kernel=eval_kernel)
        A, me, fe, g = emul.action_grad(desc, 20, XP[None, :], 2.5)
        assert abs(me[0] - meo) <= 1e-12 * abs(meo) and abs(A[0] - Ao) <= 1e-12 * abs(Ao)
        assert np.abs(g[0] - go).max() <= 1e-11 * np.abs(go).max()
