"""User-defined activations and full measurement matrices of the network action (va_nnet.py:71, 136-139, 260-264):
the NumPy oracle, the activation tracer and -- on the GPU -- the generated activation module and the kernels'
matrix contraction, against what the reference's own va_nnet.Annealer produced for a layer map that is in no
registry (z * sigmoid(z)) and for RM = [RMin, RMout] (tests/golden/nnet_extra.npz,
oracle/gen_golden_nnet.py:extra_cases)."""
import ctypes as C
import os

import numpy as np
import pytest

import va_nnet_oracle as vno
from _util import load_npz_cases
from varanneal_amd import _capi, codegen, twin

NAMES = ["g11_swish_ragged", "g11_matrix_rm_sigmoid", "g11_matrix_rm_swish_wide"]


def swish(x, W, b):
    z = np.dot(W, x) + b
    return z / (1.0 + np.exp(-z))


SWISH = (lambda z: z / (1.0 + np.exp(-z)),
         lambda a, z: 1.0 / (1.0 + np.exp(-z)) * (1.0 + z * (1.0 - 1.0 / (1.0 + np.exp(-z)))))


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("nnet_extra.npz")


def _rm(c):
    return [c["RMin"], c["RMout"]] if "RMin" in c else c["RM"]


def _oracle(c):
    act = SWISH if str(c["act"]) == "swish" else str(c["act"])
    return vno.NnetProblem(c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                           c["P"], c["Pidx"], act=act)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(gold, name):
    c = gold[name]
    pb = _oracle(c)
    A, me, fe, g = pb.action_grad(c["XP"], 1.0)
    assert abs(A - c["A"]) <= 1e-12 * abs(c["A"]) and abs(me - c["me"]) <= 1e-12 * abs(c["A"])
    assert abs(fe - c["fe"]) <= 1e-12 * abs(c["A"])
    assert np.abs(g - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()


def test_activation_tracer():
    g, dg, z = codegen.trace_activation(swish)
    import sympy as sp
    zz = np.linspace(-4, 4, 41)
    assert np.allclose(sp.lambdify(z, g, "numpy")(zz), SWISH[0](zz), rtol=1e-14)
    assert np.allclose(sp.lambdify(z, dg, "numpy")(zz), SWISH[1](None, zz), rtol=1e-13)
    text = codegen.activation_header(g, dg, z, "swish")
    assert "struct ActUser" in text and "exp(-z)" in text
    # a layer map that is not g(W.x + b) / that branches on values is refused, not mis-traced
    with pytest.raises(TypeError):
        codegen.trace_activation(lambda x, W, b: np.dot(W, x) * x[0] + b)
    with pytest.raises(TypeError):
        codegen.trace_activation(lambda x, W, b: np.dot(W, x) + b if (np.dot(W, x) + b)[0] > 0 else x)
    with pytest.raises(TypeError):
        codegen.trace_activation(lambda x, W, b: np.tanh(np.dot(W, x)) + b)      # bias outside g


def test_activation_module_cross_compiles_and_registers():
    m = codegen.activation_module_for(swish)
    assert os.path.exists(m["so"])
    L = C.CDLL(m["so"])
    v = (C.c_int * 3)()
    L.va_user_act_info(v)
    assert v[0] > 0 and hasattr(L, "va_user_act_launch")
    aid = _capi.load_act_module(m["so"])
    assert aid >= 1000 and _capi.load_act_module(m["so"]) == aid            # cached per path


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_reference(gold, name):
    c = gold[name]
    act = str(c["act"])
    if act == "swish":
        act = _capi.load_act_module(codegen.activation_module_for(swish)["so"])
    B = 3
    rng = np.random.RandomState(11)
    XP = c["XP"]
    XPb = np.stack([XP, XP + 0.01 * rng.randn(XP.size), XP])
    pr = _capi.NnetProblem(B, c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                           np.tile(c["P"], (B, 1)), c["Pidx"], act=act)
    A, me, fe, g = pr.action_grad(XPb, 1.0)
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * abs(c["A"])
        assert abs(me[b] - c["me"]) <= 1e-12 * abs(c["A"]) and abs(fe[b] - c["fe"]) <= 1e-12 * abs(c["A"])
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    A1, me1, fe1, g1 = _oracle(c).action_grad(XPb[1], 1.0)
    assert abs(A[1] - A1) <= 1e-12 * abs(A1) and np.abs(g[1] - g1).max() <= 1e-10 * np.abs(g1).max()
    # line-search points go through the same kernels with x + stp * d formed on the way
    r = pr.minimize_lbfgs(XPb, 1.0, {'gtol': 1e-10, 'ftol': 1e-12, 'maxiter': 30, 'maxfun': 200})
    assert np.all(r["A"] < A)
    pr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("structure,M", [([9, 14, 6], 12), ([70, 33, 129, 5], 300)])     # single-kernel / tiled path
def test_custom_activation_on_larger_shapes_against_oracle(structure, M):
    """a traced custom activation against the NumPy oracle to 1e-12"""
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(0, structure[-1], 2)]
    dout = dout[:, Lidx[1]]
    X0, P0, Pidx = twin.nnet_initial_guess(structure, M, 7, False)
    XP = np.append(X0.ravel(), P0[Pidx])
    aid = _capi.load_act_module(codegen.activation_module_for(swish)["so"])
    pr = _capi.NnetProblem(2, structure, din, dout, Lidx, [2.0, 3.0], 0.3, np.tile(P0, (2, 1)), Pidx, act=aid)
    A, me, fe, g = pr.action_grad(np.stack([XP, XP]), 4.0)
    pb = vno.NnetProblem(structure, din, dout, Lidx, np.array([2.0, 3.0]), 0.3, P0, Pidx, act=SWISH)
    A0, me0, fe0, g0 = pb.action_grad(XP, 4.0)
    assert abs(A[1] - A0) <= 1e-12 * abs(A0) and abs(fe[1] - fe0) <= 1e-12 * abs(A0)
    assert np.abs(g[1] - g0).max() <= 1e-12 * np.abs(g0).max() * 100        # (sums of up to 300 x 129 products)
    pr.close()


@pytest.mark.gpu
def test_custom_activation_through_the_fused_kernel():
    """the generated module's k_nnet_fb instantiation (asked for: a generated activation runs the separate kernels unasked)
    against the NumPy oracle and the separate kernels, evaluation and minimisation"""
    structure, M = [70, 33, 128, 5], 300
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(0, structure[-1], 2)]
    dout = dout[:, Lidx[1]]
    X0, P0, Pidx = twin.nnet_initial_guess(structure, M, 7, False)
    XP = np.append(X0.ravel(), P0[Pidx])[None, :]
    aid = _capi.load_act_module(codegen.activation_module_for(swish)["so"])
    opts = {'gtol': 1e-10, 'ftol': 1e-12, 'maxiter': 10, 'maxfun': 100}
    with _capi.NnetProblem(1, structure, din, dout, Lidx, 2.0, 0.3, P0[None, :], Pidx, act=aid) as pr:
        A0, me0, fe0, g0 = pr.action_grad(XP, 4.0)
        r0 = pr.minimize_lbfgs(XP, 4.0, opts)
        pr.tune(nnet_fused=1)
        A1, me1, fe1, g1 = pr.action_grad(XP, 4.0)
        r1 = pr.minimize_lbfgs(XP, 4.0, opts)
    Ao, meo, feo, go = vno.NnetProblem(structure, din, dout, Lidx, 2.0, 0.3, P0, Pidx, act=SWISH).action_grad(XP[0], 4.0)
    assert abs(A1[0] - Ao) <= 1e-12 * abs(Ao) and abs(fe1[0] - feo) <= 1e-12 * abs(Ao)
    assert np.abs(g1[0] - go).max() <= 1e-10 * np.abs(go).max()
    assert not np.array_equal(g1, g0) and np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    assert (r1["nit"][0], r1["nfev"][0]) == (r0["nit"][0], r0["nfev"][0]) and abs(r1["A"][0] - r0["A"][0]) <= 1e-9 * abs(r0["A"][0])


@pytest.mark.gpu
def test_annealer_takes_any_layer_map(gold):
    """through the drop-in: set_activation(callable) -> trace -> module; two rungs, actions decrease and the
    stored minimiser reproduces them"""
    from varanneal_amd import va_nnet
    c = gold["g11_swish_ragged"]
    a = va_nnet.Annealer()
    a.set_structure(c["structure"])
    a.set_activation(swish)
    a.set_input_data(c["din"]); a.set_output_data(c["dout"])
    M, ND = int(c["M"]), int(np.sum(c["structure"]))
    X0 = c["XP"][:M * ND].copy()                         # one seed: the flat state vector (va_nnet.py:440)
    a.anneal(X0, c["P"].copy(), 2.0, np.arange(3), c["RM"], float(c["RF0"]), list(c["Pidx"]),
             Lidx=[c["Lin"], c["Lout"]], init_to_data=False, opt_args={'gtol': 1e-8, 'ftol': 1e-10}, verbose=False)
    assert isinstance(a._act, int) and a._act >= 1000
    assert a.A_array[0] < c["A"] and np.all(a.exitflags == 0)
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    a.close()
