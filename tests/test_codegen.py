"""CPU: generated right-hand sides (varanneal_amd/codegen.py).  A user callable is traced,
differentiated and emitted as `struct RhsUser`; the generated code is (1) compiled into
the CPU emulator and run through the same flat tile phases the device module uses, and
checked against complex-step derivatives through a NumPy restatement of the reference's
action with the ORIGINAL Python callable; (2) cross-compiled with hipcc for gfx950."""
import os

import numpy as np
import pytest

import va_oracle
from cpu_emul import emul
from models.nakl import PB, l96_damped, nakl
from varanneal_amd import _capi, codegen


def _nakl_problem(N=41, seed=0):
    rng = np.random.RandomState(seed)
    D, NP = 4, 18
    t = 0.02 * np.arange(N)
    stim = 20.0 * np.sin(0.7 * t) + 5.0 * rng.randn(N)
    Y = (-60.0 + 30.0 * rng.rand(N, 1))
    X = np.column_stack([-70.0 + 60.0 * rng.rand(N), 0.2 + 0.6 * rng.rand(N, 3)])
    P = np.array([b[0] + (b[1] - b[0]) * rng.rand() for b in PB])
    RF0 = np.resize(np.array([1e-4, 1.0, 1.0, 1.0]), (N - 1, D))      # per-component RF (tutorial)
    return D, NP, N, t, stim, Y, X, P, RF0


@pytest.fixture(scope="module")
def nakl_module():
    return codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1)


def test_trace_reproduces_the_callable():
    ex, sy = codegen.trace(nakl, 4, 18, 1, 1)
    assert len(ex) == 4 and "tanh" in str(ex[1])
    codegen.check_against(nakl, ex, sy, 4, 18, 1, 1)

    def branching(t, x, p):
        return x if x[0, 0] > 0 else -x
    with pytest.raises(TypeError):
        codegen.trace(branching, 3, 1)


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_generated_nakl_matches_complex_step(nakl_module, disc):
    D, NP, N, t, stim, Y, X, P, RF0 = _nakl_problem()
    Pidx = list(range(18))
    XP = np.append(X.ravel(), P)
    rf = 1.5 ** 6
    fun = lambda z: va_oracle.numpy_action_generic(nakl, z, D, N, Y, [0], 0.02, 1.0, RF0 * rf, NP, Pidx, P,
                                                   disc, t_model=t, stim=stim)
    A0, me0, fe0 = fun(XP)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, [0], 0.02, 1.0, RF0, P[None, :], Pidx, disc=disc, rhs=1000,
                                 t_model=t, stim=stim)
    A, me, fe, g = emul.action_grad(desc, 16, XP[None, :], rf, user_header=nakl_module["header"])
    assert abs(A[0] - A0) <= 1e-12 * abs(A0) and abs(me[0] - me0) <= 1e-12 * abs(A0)
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    # partial estimation: only 3 of the 18 parameters in XP (va_ode.py:177-181)
    Pidx = [1, 7, 16]
    XPs = np.append(X.ravel(), P[Pidx])
    fun = lambda z: va_oracle.numpy_action_generic(nakl, z, D, N, Y, [0], 0.02, 1.0, RF0 * rf, NP, Pidx, P,
                                                   disc, t_model=t, stim=stim)
    desc, keep = _capi.make_desc(1, D, N, Y, [0], 0.02, 1.0, RF0, P[None, :], Pidx, disc=disc, rhs=1000,
                                 t_model=t, stim=stim)
    A, me, fe, g = emul.action_grad(desc, 16, XPs[None, :], rf, user_header=nakl_module["header"])
    gs = va_oracle.complex_step_grad(fun, XPs)
    assert abs(A[0] - fun(XPs)[0]) <= 1e-12 * abs(A0)
    assert np.abs(g[0] - gs).max() <= 1e-10 * np.abs(gs).max()


def test_generated_time_dependent_rhs():
    D, NP, N = 12, 2, 31
    m = codegen.module_for(l96_damped, D, NP, compile=False)          # (the emulator compiles the header itself)
    rng = np.random.RandomState(4)
    t = 0.025 * np.arange(N)
    Y = rng.randn(N, 5); Lidx = [0, 2, 5, 7, 10]
    P = np.array([8.0, 1.1])
    XP = np.append(3.0 * rng.randn(N * D), P)
    fun = lambda z: va_oracle.numpy_action_generic(l96_damped, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [0, 1], P,
                                                   "trapezoid", t_model=t)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None, :], [0, 1], rhs=1000, t_model=t)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()


def test_module_cross_compiles_for_gfx950(nakl_module):
    assert os.path.exists(nakl_module["so"]) and os.path.getsize(nakl_module["so"]) > 10000
    import ctypes as C
    L = C.CDLL(nakl_module["so"])               # loads without a GPU; exports the module ABI
    v = (C.c_int * 5)()
    L.va_user_rhs_info(v)
    assert list(v)[:3] == [18, 4, 1]
    assert hasattr(L, "va_user_launch_eval")


def _golden_nakl():
    from _util import load_npz_cases
    return load_npz_cases("nakl.npz")


@pytest.mark.parametrize("name", ["g5_nakl_SimpsonHermite_rf1e+00", "g5_nakl_SimpsonHermite_rf5e+01",
                                  "g5_nakl_trapezoid_rf1e+00", "g5_nakl_trapezoid_rf2e+03"])
def test_generated_nakl_matches_reference_golden(nakl_module, name):
    """A, measurement/model split and gradient the reference itself produced for the tutorial's
    NaKL model (oracle/gen_golden.py:nakl_cases) -- pins both the NumPy restatement for generic
    f and the generated code run through the emulator."""
    c = _golden_nakl()[name]
    D, N = int(c["D"]), int(c["N_model"])
    RF0 = np.resize(c["RF0"], (N - 1, D))
    XP = c["XP"]
    P = XP[N * D:]
    Pidx = list(range(18))
    rf = float(c["rf_scale"])
    A0, me0, fe0 = va_oracle.numpy_action_generic(nakl, XP, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]),
                                                  RF0 * rf, 18, Pidx, P, str(c["disc"]), t_model=c["t"],
                                                  stim=c["stim"])
    assert abs(A0 - c["A"]) <= 1e-12 * c["A"] and abs(me0 - c["me"]) <= 1e-12 * c["A"]
    desc, keep = _capi.make_desc(1, D, N, c["Y"], [0], float(c["dt_model"]), float(c["RM"]), RF0, P[None, :], Pidx,
                                 disc=str(c["disc"]), rhs=1000, t_model=c["t"], stim=c["stim"])
    A, me, fe, g = emul.action_grad(desc, 16, XP[None, :], rf, user_header=nakl_module["header"])
    assert abs(A[0] - c["A"]) <= 1e-12 * c["A"]
    assert abs(me[0] - c["me"]) <= 1e-12 * c["A"] and abs(fe[0] - c["fe"]) <= 1e-12 * c["A"]
    assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()


# ---- the column form (codegen.column_form): the same models on the wave-private column-run kernel -------------
def _l96_plain(t, x, p):
    """Lorenz-96 written by a user (NOT the registry's callable): must be recognised as a translation-invariant
    stencil and come out with the built-in's structure (3 neighbour columns, 2 exchanged products)."""
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - x + p[0]


def test_column_form_structure():
    ex, sy = codegen.trace(_l96_plain, 20, 1)
    c = codegen.column_form(ex, sy, 20, 1, 0)
    assert c["uniform"] and c["offsets"] == [-2, -1, 1] and c["NE"] == 2       # as RhsL96s (csrc/va_tile4.h)
    ex, sy = codegen.trace(nakl, 4, 18, 1, 1)
    c = codegen.column_form(ex, sy, 4, 18, 1)
    assert not c["uniform"] and c["offsets"] == [1, 2, 3] and c["NE"] == 3      # dense: every other column
    ex, sy = codegen.trace(l96_damped, 12, 2)
    assert "USES_T = true" in codegen.column_form(ex, sy, 12, 2, 0)["text"]

    def coupled9(t, x, p):                       # neither translation-invariant nor small: flat kernel only
        return np.cumsum(x, 1) * p[0]
    ex, sy = codegen.trace(coupled9, 9, 1)
    assert codegen.column_form(ex, sy, 9, 1, 0) is None


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_column_form_nakl_matches_flat_form_and_complex_step(disc):
    """dense column form (switch on the column) + stimulus + per-component RF + 18 estimated parameters"""
    D, NP, N, t, stim, Y, X, P, RF0 = _nakl_problem(N=61)
    m = codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1, col_variant=lambda ne, gh: (4, _capi.DISC[disc], 4, 0),
                           compile=False)
    assert m["col"] is not None and "#define VA_USER_COL" in m["text"]
    Pidx = list(range(18))
    XP = np.append(X.ravel(), P)
    rf = 1.5 ** 6
    fun = lambda z: va_oracle.numpy_action_generic(nakl, z, D, N, Y, [0], 0.02, 1.0, RF0 * rf, NP, Pidx, P,
                                                   disc, t_model=t, stim=stim)
    g0 = va_oracle.complex_step_grad(fun, XP)
    out = {}
    for ek in (1, 4):
        desc, keep = _capi.make_desc(1, D, N, Y, [0], 0.02, 1.0, RF0, P[None, :], Pidx, disc=disc, rhs=1000,
                                     t_model=t, stim=stim, eval_kernel=ek)
        out[ek] = emul.action_grad(desc, 64, XP[None, :], rf, user_header=m["header"])
    A0 = fun(XP)[0]
    for ek in (1, 4):
        A, me, fe, g = out[ek]
        assert abs(A[0] - A0) <= 1e-12 * abs(A0)
        assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    assert np.abs(out[4][3] - out[1][3]).max() <= 1e-12 * np.abs(g0).max()


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite"])
def test_column_form_traced_l96_equals_builtin(disc):
    """the traced Lorenz-96 on the column-run phases against the built-in on the same phases"""
    D, N = 20, 85
    rng = np.random.RandomState(2)
    Lidx = [0, 3, 6, 9, 12, 15, 18]
    Y = rng.randn(N, len(Lidx)); P = np.array([8.17])
    XP = np.append(3.0 * rng.randn(N * D), P)
    m = codegen.module_for(_l96_plain, D, 1, col_variant=lambda ne, gh: (4, _capi.DISC[disc], 6, 1), compile=False)
    res = {}
    for rhs in ("lorenz96", 1000):
        desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None, :], [0], disc=disc, rhs=rhs,
                                     eval_kernel=4)
        res[rhs] = emul.action_grad(desc, 72, XP[None, :], 8.0, user_header=m["header"])
    A, me, fe, g = res[1000]
    Ab, meb, feb, gb = res["lorenz96"]
    assert abs(A[0] - Ab[0]) <= 1e-13 * abs(Ab[0]) and abs(fe[0] - feb[0]) <= 1e-13 * abs(Ab[0])
    assert np.abs(g - gb).max() <= 1e-13 * np.abs(gb).max()


def test_column_form_time_dependent_stencil():
    D, NP, N = 12, 2, 45
    m = codegen.module_for(l96_damped, D, NP, col_variant=lambda ne, gh: (4, 1, 4, 1), compile=False)
    assert m["col"]["uniform"]
    rng = np.random.RandomState(4)
    t = 0.025 * np.arange(N)
    Y = rng.randn(N, 5); Lidx = [0, 2, 5, 7, 10]
    P = np.array([8.0, 1.1])
    XP = np.append(3.0 * rng.randn(N * D), P)
    fun = lambda z: va_oracle.numpy_action_generic(l96_damped, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, [0, 1], P,
                                                   "trapezoid", t_model=t)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None, :], [0, 1], rhs=1000, t_model=t,
                                 eval_kernel=4)
    A, me, fe, g = emul.action_grad(desc, 80, XP[None, :], 1.0, user_header=m["header"])
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()


def test_column_module_exports_its_variant():
    import ctypes as C
    m = codegen.module_for(_l96_plain, 20, 1, col_variant=lambda ne, gh: (4, 1, 7, 1) if ne == 2 else None)
    L = C.CDLL(m["so"])
    v = (C.c_int * 12)()
    L.va_user_variant_info(v)
    assert list(v)[:6] == [4, 1, 7, 1, 2, 0] and list(v)[10] == 0 and hasattr(L, "va_user_launch_variant") and hasattr(L, "va_user_prepare_variant")


def test_module_without_variant_reports_none(nakl_module):
    import ctypes as C
    L0 = C.CDLL(nakl_module["so"])                              # no variant asked for: flat kernel only
    v = (C.c_int * 12)()
    L0.va_user_variant_info(v)
    assert list(v)[0] == 0 and not hasattr(L0, "va_user_launch_variant")


def test_eval_plan_matches_the_geometry_rules():
    assert _capi.eval_plan(64, 20, 1000, "trapezoid", 2, 2) == (4, 1, 7, 1)          # C3: 12 tiles x 64 = 3 per CU
    assert _capi.eval_plan(512, 20, 1000, "trapezoid", 2, 2) == (4, 1, 7, 1)         # many rounds: fewest staged rows
    assert _capi.eval_plan(1, 4, 2001, "SimpsonHermite", 3, rf_array=True) == (4, 2, 4, 0)
    assert _capi.eval_plan(1, 4, 2001, "SimpsonHermite", 3, bounded=True) is None   # bounds: flat kernel
    assert _capi.eval_plan(1, 22, 200, "trapezoid", 2) is None                 # D = 22: two runs leave 20 lanes idle...
    assert _capi.eval_plan(1, 22, 200, "trapezoid", 2, 2)[0] == 3              # ...the workgroup kernel takes it
    assert _capi.eval_plan(1, 20, 200, "trapezoid", 2, 2, p_time_dependent=True) is None
    assert _capi.eval_plan(64, 200, 5000, "trapezoid", 0, 2) == (3, 1, 8, 256)  # C4's shape: one lane per column
    assert _capi.eval_plan(64, 200, 5000, "trapezoid", 2, 0) is None
    # instantiations that exist for the built-in right-hand side only: Simpson-Hermite at the C3 shape in runs of 12 rows (one
    # round of resident workgroups), merr_nskip on the scalar-weight kernel's run length (row mask instead of weight registers)
    assert _capi.eval_plan(64, 20, 1001, "SimpsonHermite", 2, 2, builtin=True) == (4, 2, 12, 1)
    assert _capi.eval_plan(64, 20, 1001, "SimpsonHermite", 2, 2) == (4, 2, 4, 1)
    assert _capi.eval_plan(32, 20, 1001, "SimpsonHermite", 2, 2, builtin=True) == (4, 2, 4, 1)       # already one round
    assert _capi.eval_plan(64, 20, 1001, "trapezoid", 2, 2, merr_nskip=2, builtin=True) == (4, 1, 7, 0)
    assert _capi.eval_plan(64, 20, 1001, "trapezoid", 2, 2, merr_nskip=2) == (4, 1, 5, 0)
    assert _capi.eval_plan(64, 20, 1001, "SimpsonHermite", 2, 2, merr_nskip=2, builtin=True) == (4, 2, 12, 0)
    assert _capi.eval_plan(64, 20, 1000, "trapezoid", 2, 2, rf_array=True, builtin=True) == (4, 1, 7, 0)    # (weights parked in LDS)
    assert _capi.eval_plan(64, 20, 1000, "trapezoid", 2, 2, rf_array=True) == (4, 1, 5, 0)
    assert _capi.eval_plan(64, 2000, 500, "trapezoid", 0, 2) is None            # beyond 1024 columns: flat


# ---- the ghosted form (codegen.ghost_form): wide stencils on the workgroup column-run kernel ------------------
def _wide_stencil(t, x, p):
    """not Lorenz-96: a five-point stencil with a cubic term and two parameters"""
    return (np.roll(x, 1, 1) * (np.roll(x, -2, 1) - np.roll(x, 2, 1)) - p[1] * x ** 3
            + p[0] * np.roll(x, -1, 1))


def test_ghost_form_structure():
    ex, sy = codegen.trace(_l96_plain, 100, 1)
    g = codegen.ghost_form(ex, sy, 100, 1, 0)
    assert g["GHOST"] == 2 and g["offsets"] == [-2, -1, 0, 1]                  # as RhsL96g (csrc/va_tile3.h)
    ex, sy = codegen.trace(_wide_stencil, 70, 2)
    g = codegen.ghost_form(ex, sy, 70, 2, 0)
    assert g["GHOST"] == 4                       # the adjoint of x_{j-1} x_{j+2} reads three columns away
    ex, sy = codegen.trace(l96_damped, 12, 2)
    assert codegen.ghost_form(ex, sy, 12, 2, 0) is None                        # explicit time: column form only
    ex, sy = codegen.trace(nakl, 4, 18, 1, 1)
    assert codegen.ghost_form(ex, sy, 4, 18, 1) is None


@pytest.mark.parametrize("f,D,NP", [(_l96_plain, 72, 1), (_wide_stencil, 70, 2)])
@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_ghost_form_matches_flat_form_and_complex_step(f, D, NP, disc):
    N = 41
    rng = np.random.RandomState(D)
    Lidx = list(range(0, D, 5))
    Y = rng.randn(N, len(Lidx)); P = np.array([8.0, 0.05][:NP])
    XP = np.append(2.0 * rng.randn(N * D), P)
    m = codegen.module_for(f, D, NP, col_variant=lambda ne, gh: (3, _capi.DISC[disc], 4, 128) if gh else None,
                           compile=False)
    assert m["ghost"] is not None and "#define VA_USER_GHOST" in m["text"]
    Pidx = list(range(NP))
    fun = lambda z: va_oracle.numpy_action_generic(f, z, D, N, Y, Lidx, 0.025, 4.0, 0.3, NP, Pidx, P, disc)
    g0 = va_oracle.complex_step_grad(fun, XP)
    out = {}
    for ek in (1, 3):
        desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, P[None, :], Pidx, disc=disc, rhs=1000,
                                     eval_kernel=ek)
        out[ek] = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
    A0 = fun(XP)[0]
    for ek in (1, 3):
        A, me, fe, g = out[ek]
        assert abs(A[0] - A0) <= 1e-12 * abs(A0)
        assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    assert np.abs(out[3][3] - out[1][3]).max() <= 1e-12 * np.abs(g0).max()


def test_ghost_module_exports_its_variant():
    import ctypes as C
    m = codegen.module_for(_l96_plain, 200, 1, col_variant=lambda ne, gh: _capi.eval_plan(64, 200, 5000, "trapezoid", ne, gh))
    assert m["col_variant"] == (3, 1, 8, 256) and m["col"] is None
    L = C.CDLL(m["so"])
    v = (C.c_int * 12)()
    L.va_user_variant_info(v)
    assert list(v)[:6] == [3, 1, 8, 256, 0, 2]


def test_gather_coefficients_keep_their_digits():
    """proportional products share a slot with a constant factor: the factor is printed from its value (2.1 and 11
    used to lose their leading digits to a text replacement of "1 * ")"""
    def f(t, x, p):
        return np.roll(x, 1, 1) + 2.1 * np.roll(x, -1, 1) - p[0] * x

    def g(t, x, p):
        return np.roll(x, 1, 1) + 11 * np.roll(x, -1, 1) - p[0] * x * x

    def h(t, x, p):
        return np.roll(x, 1, 1) - np.roll(x, -1, 1) - p[0] * x
    for fun, want in ((f, "r[0] + (2.1000000000000001) * r[1]"), (g, "r[0] + (11) * r[1]"), (h, "r[0] + -r[1]")):
        m = codegen.module_for(fun, 20, 1, col_variant=lambda ne, gh: (4, 1, 6, 1), compile=False)
        line = [ln for ln in m["text"].splitlines() if "double gather(" in ln][0]
        assert want in line, line


def test_wide_stencil_gets_the_streaming_variant():
    """a three-argument callback is handed the column form's reaches; with them (and the observed columns) the plan
    for BASELINE config 4's shape is the streaming kernel, and the header carries the column form"""
    Lidx = list(range(0, 200, 5))
    assert _capi.eval_plan(64, 200, 5000, "trapezoid", 2, 2, reach=(2, 1, 1, 2), Lidx=Lidx) == (5, 1, 0, 0)
    assert _capi.eval_plan(64, 200, 5001, "SimpsonHermite", 2, 2, reach=(2, 1, 1, 2), Lidx=Lidx) == (5, 2, 0, 0)
    assert _capi.eval_plan(64, 200, 5000, "trapezoid", 2, 2, reach=(2, 1, 1, 2), Lidx=Lidx[:-1])[0] == 5      # (odd L: a pad column on the device)
    assert _capi.eval_plan(64, 201, 5000, "trapezoid", 2, 2, reach=(2, 1, 1, 2), Lidx=Lidx)[0] == 3            # odd D: rows are not 16-byte aligned
    assert _capi.eval_plan(64, 200, 5000, "trapezoid", 2, 2, reach=(2, 1, 1, 2), Lidx=Lidx, rf_array=True)[0] == 5      # (per-row weights stream too)
    seen = {}

    def cb(ne, gh, reach):
        seen["reach"] = reach
        return _capi.eval_plan(64, 200, 5000, "trapezoid", ne, gh, reach=reach, Lidx=Lidx)
    m = codegen.module_for(_l96_plain, 200, 1, col_variant=cb, compile=False)
    assert seen["reach"] == (2, 1, 1, 2) and m["col_variant"] == (5, 1, 0, 0)
    assert "struct RhsUserCol" in m["text"] and "struct RhsUserG" not in m["text"]


def _ring_of_units(t, x, p):
    """NOT a stencil: a ring of five identical 4-state units (D = 20).  Unit u has a fast variable v = x[4u] driven by
    its own three slow variables and diffusively coupled to the neighbouring units' v; the slow variables relax towards
    polynomial functions of v.  Three parameters shared by all units."""
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


def test_ring_of_identical_units_gets_a_periodic_column_form():
    """block-periodic models (period P > 1): P column classes, one body each (`switch (col % P)`), the neighbour offsets
    their union -- instead of the flat kernel's switch over all D columns"""
    m = codegen.module_for(_ring_of_units, 20, 3, col_variant=lambda ne, gh: (4, 1, 5, 1), compile=False)
    c = m["col"]
    assert c is not None and not c["uniform"] and c["period"] == 4
    assert c["offsets"] == [-4, -3, -2, -1, 1, 2, 3, 4] and c["NB"] == 8
    assert "switch (col % 4)" in m["text"] and "case 3:" in m["text"] and "case 4:" not in m["text"].split("struct RhsUserCol")[1].split("struct")[0]
    # a model without any such structure (every unit different) has no column form at D = 20
    def irregular(t, x, p):
        return np.stack([x[..., (i + 1) % 20] * (1.0 + 0.1 * i) - x[..., i] ** 2 + p[0] for i in range(20)], axis=-1)
    m2 = codegen.module_for(irregular, 20, 1, col_variant=lambda ne, gh: (4, 1, 5, 1), compile=False)
    assert m2["col"] is None


# ---- dense constant linear part (codegen.linear_split) and the switch-free flat form --------------------------------
def _coupled20():
    C = np.random.RandomState(0).randn(20, 20) / np.sqrt(20.0)

    def coupled(t, x, p):
        return x @ C.T - p[1] * x ** 3 + p[0]
    return coupled, C


def test_linear_split_takes_the_dense_constant_part():
    f, C = _coupled20()
    ex, sy = codegen.trace(f, 20, 2)
    A0, rest = codegen.linear_split(ex, sy, 20)
    assert np.array_equal(A0, C)
    assert all(not r.has(sy["x"][(i + 1) % 20]) for i, r in enumerate(rest))     # what is left is local
    # Lorenz-96: one constant linear entry per row (-x_i) -- not worth a matrix product; narrow states never are
    ex, sy = codegen.trace(_l96_plain, 20, 1)
    assert codegen.linear_split(ex, sy, 20) is None
    ex, sy = codegen.trace(lambda t, x, p: x @ np.ones((8, 8)) * 0.5 + p[0], 8, 1)
    assert codegen.linear_split(ex, sy, 8) is None
    # a coefficient that depends on a parameter stays with the element-wise code
    ex, sy = codegen.trace(lambda t, x, p: p[0] * (x @ C.T), 20, 1)
    assert codegen.linear_split(ex, sy, 20) is None


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_linear_module_matches_complex_step(disc):
    """the struct of a split model describes the rest only (translation-invariant here: no switch) and carries the
    tables of A0; the emulator adds the linear part with plain loops where the device uses the matrix cores"""
    f, C = _coupled20()
    D, NP, N = 20, 2, 33
    m = codegen.module_for(f, D, NP, compile=False)
    assert m["lin"] is not None and "LINEAR = true" in m["text"] and "switch (i)" not in m["text"] and "va_lin_A0T" in m["text"]
    rng = np.random.RandomState(0)
    Y = rng.randn(N, 6); Lidx = [0, 3, 5, 9, 12, 17]
    P = np.array([1.5, 0.3])
    XP = np.append(rng.randn(N * D), P)
    fun = lambda z: va_oracle.numpy_action_generic(f, z, D, N, Y, Lidx, 0.02, 2.0, 0.7, NP, [0, 1], P, disc)
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.02, 2.0, 0.7, P[None, :], [0, 1], disc=disc, rhs=1000)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
    assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()


def test_translation_invariant_models_get_a_switch_free_flat_form():
    """the flat kernel's struct of a stencil: one body with cyclic index arithmetic (a per-component switch diverges
    64 ways in a wave); checked through the emulator against the built-in Lorenz-96 of the oracle"""
    D, N = 20, 31
    m = codegen.module_for(_l96_plain, D, 1, compile=False)
    assert "switch (i)" not in m["text"] and "x[w(i - 2)]" in m["text"]
    rng = np.random.RandomState(1)
    Y = rng.randn(N, 4); Lidx = [0, 5, 11, 19]
    XP = np.append(3.0 * rng.randn(N * D), 8.1)
    pb = va_oracle.Problem(D, N, Y, Lidx, 0.025, 4.0, 0.3, np.array([8.1]), [0], disc="trapezoid")
    A0, me0, fe0, g0 = pb.action_grad(XP, 1.0)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.3, np.array([[8.1]]), [0], disc="trapezoid", rhs=1000)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
    assert abs(A[0] - A0) <= 1e-12 * abs(A0) and np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
    # a model that is not translation-invariant keeps the switch
    assert "switch (i)" in codegen.module_for(nakl, 4, 18, nstim=1, stim_ndim=1, compile=False)["text"]


def test_linear_module_with_stimulus_and_explicit_time():
    """a dense linear part next to a rest that reads the stimulus and the time (f(t, x, (p, stim)), va_ode.py:345-354):
    the split leaves both in the element-wise code -- switch-free, they are the same for every component -- and the
    emulator's gradient matches complex-step"""
    D, NP, N = 16, 2, 25
    C = np.random.RandomState(3).randn(D, D) / 4.0

    def driven(t, x, ps):
        p, stim = ps
        drive = p[0] * stim * np.cos(0.3 * t)
        return x @ C.T - p[1] * x ** 3 + drive[:, None]
    m = codegen.module_for(driven, D, NP, nstim=1, stim_ndim=1, compile=False)
    assert m["lin"] is not None and np.array_equal(m["lin"], C) and "switch (i)" not in m["text"]
    rng = np.random.RandomState(0)
    t = 0.05 * np.arange(N)
    stim = np.sin(1.3 * t) + 0.1 * rng.randn(N)
    Y = rng.randn(N, 4); Lidx = [1, 4, 9, 15]
    P = np.array([0.8, 0.2])
    XP = np.append(rng.randn(N * D), P)
    for disc in ("trapezoid", "SimpsonHermite"):
        fun = lambda z: va_oracle.numpy_action_generic(driven, z, D, N, Y, Lidx, 0.05, 2.0, 0.5, NP, [0, 1], P, disc, t_model=t, stim=stim)
        g0 = va_oracle.complex_step_grad(fun, XP)
        desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.05, 2.0, 0.5, P[None, :], [0, 1], disc=disc, rhs=1000, t_model=t, stim=stim)
        A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], 1.0, user_header=m["header"])
        assert abs(A[0] - fun(XP)[0]) <= 1e-12 * abs(A[0])
        assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()


def _relu_ring(t, x, p):
    """a piecewise model: rectified coupling, two leak rates chosen by the state, a saturating self-term
    (the reference tapes whatever executes, _autodiffmin.py:41-44; here np.maximum / np.where / np.clip trace to selects)"""
    drive = np.maximum(np.roll(x, 1, 1), 0.0)
    leak = np.where(x > 0.5, p[1], 0.5 * p[1])
    return -leak * x + p[0] * drive + 0.3 * np.clip(np.roll(x, -1, 1), -1.0, 1.5) - 0.2 * np.minimum(x, np.roll(x, 2, 1))


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_piecewise_model_traces_to_selects_and_matches_complex_step(disc):
    D, NP, N = 10, 2, 41
    ex, sy = codegen.trace(_relu_ring, D, NP)
    assert "Piecewise" in str(ex[0])
    codegen.check_against(_relu_ring, ex, sy, D, NP, 0)
    m = codegen.module_for(_relu_ring, D, NP, compile=False)
    rng = np.random.RandomState(7)
    Y = rng.randn(N, 4); Lidx = [0, 3, 5, 8]
    P = np.array([1.3, 0.8]); Pidx = [0, 1]
    XP = np.append(1.5 * rng.randn(N * D), P)
    rf = 30.0
    fun = lambda z: va_oracle.numpy_action_generic(_relu_ring, z, D, N, Y, Lidx, 0.05, 2.0, 0.1 * rf, NP, Pidx, P, disc)
    A0, me0, fe0 = fun(XP)
    g0 = va_oracle.complex_step_grad(fun, XP)          # (the derivative of the taken branch: what a tape gives)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.05, 2.0, 0.1, P[None, :], Pidx, disc=disc, rhs=1000)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], rf, user_header=m["header"])
    assert abs(A[0] - A0) <= 1e-12 * abs(A0)
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()

    def py_branch(t, x, p):                              # Python-level branching stays untraceable, and says what to write instead
        return np.array([[v if v > 0 else 0.0 * v for v in x[0]]])
    with pytest.raises(TypeError, match="np.where"):
        codegen.trace(py_branch, 3, 1)


def _many_parameters(t, x, p):
    """40 parameters (the tuned kernels and the partial-sum rows carry 24; the reference has no cap,
    varanneal/va_ode.py:564-578): every state has its own forcing p[i] and its own damping p[20 + i]"""
    D = x.shape[1]
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - p[D:2 * D] * x + p[:D]


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite"])
def test_forty_parameters_through_the_flat_kernel(disc):
    D, NP, N = 20, 40, 31
    m = codegen.module_for(_many_parameters, D, NP, compile=False,
                           col_variant=lambda ne, gh: _capi.eval_plan(1, D, N, disc, ne, gh))
    assert m["col"] is None and m["ghost"] is None            # (column forms stop at 24 parameters)
    rng = np.random.RandomState(9)
    Y = rng.randn(N, 6); Lidx = [0, 3, 7, 11, 14, 18]
    P = np.append(8.0 + rng.rand(D), 0.8 + 0.4 * rng.rand(D))
    Pidx = [0, 5, 19, 20, 23, 24, 25, 31, 39]                 # estimated ones on both sides of the 24th
    XP = np.append(3.0 * rng.randn(N * D), P[Pidx])
    rf = 20.0
    fun = lambda z: va_oracle.numpy_action_generic(_many_parameters, z, D, N, Y, Lidx, 0.025, 4.0, 0.01 * rf, NP, Pidx, P, disc)
    A0 = fun(XP)[0]
    g0 = va_oracle.complex_step_grad(fun, XP)
    desc, keep = _capi.make_desc(1, D, N, Y, Lidx, 0.025, 4.0, 0.01, P[None, :], Pidx, disc=disc, rhs=1000)
    A, me, fe, g = emul.action_grad(desc, 8, XP[None, :], rf, user_header=m["header"])
    assert abs(A[0] - A0) <= 1e-12 * abs(A0)
    assert np.abs(g[0] - g0).max() <= 1e-10 * np.abs(g0).max()
