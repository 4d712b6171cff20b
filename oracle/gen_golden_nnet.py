"""Generate tests/golden/nnet.npz from the REFERENCE's va_nnet.Annealer (TEST INFRASTRUCTURE;
runs only in the build container where /root/reference is mounted; see oracle/_refload.py).

  g6_*   single evaluations: (XP, data, structure, RM, RF) -> A, me, fe through the reference's
         A_gaussian (va_nnet.py:111-255) and the complex-step gradient through the same A
  g7_*   ladders through the reference's own anneal()/anneal_step() (va_nnet.py:267-523) +
         SciPy L-BFGS-B, with adolc.function -> reference A and adolc.gradient -> the oracle
         adjoint (oracle/va_nnet_oracle.py, checked against the complex-step goldens)
Activations are passed to the reference as Python callables f(x, W, b), as its example does
(examples/nnet_twin/nnet_twin_anneal.py:20-22).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]
import _refload                      # noqa: E402
import va_nnet_oracle as vno         # noqa: E402
from varanneal_amd import twin       # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
ACT_F = {
    "sigmoid": twin.sigmoid,
    "tanh": lambda x, W, b: np.tanh(np.dot(W, x) + b),
    "linear": lambda x, W, b: np.dot(W, x) + b,
}


def ref_annealer(vn, structure, din, dout, act):
    a = vn.Annealer()
    a.set_structure(np.asarray(structure, dtype=int))
    a.set_activation(ACT_F[act])
    a.set_input_data(din)
    a.set_output_data(dout)
    return a


def twin_rf0(structure, RM):
    """nnet_twin_anneal.py:45"""
    s = np.asarray(structure)
    return 1.0e-8 * RM * float(np.sum(s) - s[0]) / float(s[0] + s[-1])


def single_cases(vn, mnist_only=False):
    out = {}

    def run(name, structure, M, act, RM, rf_scale, weights_only, init_to_data, Lidx=None, seed=0, rf0=None):
        structure = np.asarray(structure, dtype=int)
        din_full, dout_full, _ = twin.make_nnet_twin(structure, M)
        if Lidx is None:
            Lidx = [np.arange(structure[0]), np.arange(structure[-1])]
        Lidx = [np.asarray(Lidx[0], dtype=int), np.asarray(Lidx[1], dtype=int)]
        din, dout = din_full[:, Lidx[0]], dout_full[:, Lidx[1]]
        X0, P0, Pidx = twin.nnet_initial_guess(structure, M, seed, weights_only)
        rm_scalar = float(np.mean(RM))
        RF0 = twin_rf0(structure, rm_scalar) if rf0 is None else rf0
        a = ref_annealer(vn, structure, din, dout, act)
        RMa = np.array(RM, dtype=np.float64) if np.ndim(RM) else float(RM)
        a.anneal_init(X0, P0.copy(), 1.0, np.array([0]), RMa, RF0 * rf_scale, Pidx, Lidx=Lidx,
                      init_to_data=init_to_data, disc='forwardmap', method='L-BFGS-B')
        XP = np.append(a.minpaths[0][:a.NDens], a.minpaths[0][a.NDens:][Pidx])
        A = float(a.A(XP)); me = float(a.me_gaussian(XP)); fe = float(a.fe_gaussian(XP))
        grad = _refload.complex_step_grad(a.A, XP)
        out[name] = dict(structure=structure, M=M, act=act, din=din, dout=dout, Lin=Lidx[0], Lout=Lidx[1],
                         RM=np.asarray(RM, dtype=np.float64), RF0=RF0, rf_scale=rf_scale, P=P0,
                         Pidx=np.asarray(Pidx, dtype=int), XP=XP, A=A, me=me, fe=fe, grad=grad)
        print("%-40s A=%.16e me=%.3e fe=%.3e |g|max=%.3e" % (name, A, me, fe, np.abs(grad).max()))

    if mnist_only:
        # the tutorial's MNIST network in full (VarAnneal_tutorial.ipynb:3413-3415, 3449-3450: 784-30-10, M = 2, weights
        # estimated, RM = 1, RF0 by the tutorial's formula), a third of the way up its ladder (alpha = 1.1, beta = 145);
        # synthetic twin data of that shape (the MNIST files are not shipped); complex-step gradient over all 25,468 unknowns
        run("g6_mnist_784_30_10", [784, 30, 10], 2, "sigmoid", 1.0, 1.1 ** 145, True, True, seed=6)
        return out
    tw = twin.nnet_structure(20, 10, 10, 10)
    # BASELINE C5 shape: structure [10]*20, M=2, weights estimated, biases fixed at 0
    run("g6_twin_rf1", tw, 2, "sigmoid", 1.0 / 0.005 ** 2, 1.0, True, True)
    run("g6_twin_rf1e6", tw, 2, "sigmoid", 1.0 / 0.005 ** 2, 1.0e6, True, True)
    run("g6_twin_noinit_rf1e4", tw, 2, "sigmoid", 1.0 / 0.005 ** 2, 1.0e4, True, False, seed=3)
    # ragged layers, every parameter estimated, RM = [RM_in, RM_out], partially observed ends
    run("g6_ragged_full_rm2", [12, 7, 9, 5], 5, "sigmoid", [3.0, 5.0], 1.0, False, False,
        Lidx=[[0, 1, 4, 7, 11], [0, 2, 3]], rf0=0.02)
    run("g6_ragged_tanh", [6, 33, 4], 19, "tanh", 2.0, 1.0, False, False, rf0=0.5, seed=2)
    run("g6_linear_wide", [40, 17, 3], 35, "linear", 1.0, 1.0, True, True, rf0=0.1, seed=4)
    # tutorial MNIST shape 784-30-10 (VarAnneal_tutorial.ipynb:3413-3415), M=2, value only is cheap;
    # gradient by complex step over 25k unknowns is too slow -> smaller 100-30-10 with M=4
    run("g6_mnistlike_100_30_10", [100, 30, 10], 4, "sigmoid", 1.0, 1.0e3, False, True, seed=5)
    return out


def swish(x, W, b):
    """a layer map that is in nobody's registry: z * sigmoid(z) of z = W.x + b"""
    z = np.dot(W, x) + b
    return z / (1.0 + np.exp(-z))


def extra_cases(vn):
    """g11_*: a user-defined activation (the reference takes any callable, va_nnet.py:71) and full measurement
    matrices RM = [RMin, RMout] (va_nnet.py:136-139; not symmetric on purpose)."""
    out = {}
    ACT_F["swish"] = swish

    def run(name, structure, M, act, RM, Lidx, rf0, seed):
        structure = np.asarray(structure, dtype=int)
        din_full, dout_full, _ = twin.make_nnet_twin(structure, M)
        Lidx = [np.asarray(Lidx[0], dtype=int), np.asarray(Lidx[1], dtype=int)]
        din, dout = din_full[:, Lidx[0]], dout_full[:, Lidx[1]]
        X0, P0, Pidx = twin.nnet_initial_guess(structure, M, seed, False)
        a = ref_annealer(vn, structure, din, dout, act)
        # (the reference reaches its matrix branch only for an array whose shape is not (2,): the two
        # matrices must have the same size, i.e. as many observed inputs as outputs)
        RMa = np.array(RM, dtype=np.float64)
        a.anneal_init(X0, P0.copy(), 1.0, np.array([0]), RMa, rf0, Pidx, Lidx=Lidx,
                      init_to_data=False, disc='forwardmap', method='L-BFGS-B')
        XP = np.append(a.minpaths[0][:a.NDens], a.minpaths[0][a.NDens:][Pidx])
        A = float(a.A(XP)); me = float(a.me_gaussian(XP)); fe = float(a.fe_gaussian(XP))
        grad = _refload.complex_step_grad(a.A, XP)
        rec = dict(structure=structure, M=M, act=act, din=din, dout=dout, Lin=Lidx[0], Lout=Lidx[1], RF0=rf0,
                   rf_scale=1.0, P=P0, Pidx=np.asarray(Pidx, dtype=int), XP=XP, A=A, me=me, fe=fe, grad=grad)
        if np.ndim(RM[0]) == 2:
            rec["RMin"], rec["RMout"] = np.asarray(RM[0]), np.asarray(RM[1])
        else:
            rec["RM"] = np.asarray(RM, dtype=np.float64)
        out[name] = rec
        print("%-40s A=%.16e me=%.3e fe=%.3e |g|max=%.3e" % (name, A, me, fe, np.abs(grad).max()))

    rng = np.random.RandomState(11)
    run("g11_swish_ragged", [12, 7, 9, 5], 5, "swish", [3.0, 5.0], [[0, 1, 4, 7, 11], [0, 2, 3]], 0.02, 1)
    Rin = 3.0 * np.eye(4) + 0.4 * rng.randn(4, 4); Rout = 5.0 * np.eye(4) + 0.4 * rng.randn(4, 4)
    run("g11_matrix_rm_sigmoid", [12, 7, 9, 5], 5, "sigmoid", [Rin, Rout], [[0, 4, 7, 11], [0, 2, 3, 4]], 0.02, 2)
    run("g11_matrix_rm_swish_wide", [40, 70, 6], 37, "swish", [4.0 * np.eye(6) + 0.2 * rng.randn(6, 6),
                                                                 2.0 * np.eye(6) + 0.2 * rng.randn(6, 6)],
        [np.arange(3, 40, 7), np.arange(6)], 0.1, 3)
    return out


def ladder_case(vn, name, structure, M, act, alpha, betas, seed=0):
    import adolc
    structure = np.asarray(structure, dtype=int)
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(structure[-1])]
    X0, P0, Pidx = twin.nnet_initial_guess(structure, M, seed, True)
    RM = 1.0 / 0.005 ** 2
    RF0 = twin_rf0(structure, RM)
    a = ref_annealer(vn, structure, din, dout, act)
    pb = vno.NnetProblem(structure, din, dout, Lidx, RM, 1.0, P0, Pidx, act=act)
    counts = dict(f=0, g=0)

    def fn(_id, XP):
        counts["f"] += 1
        return a.A(np.asarray(XP, dtype=np.float64))

    def gr(_id, XP):
        counts["g"] += 1
        pb.RF0 = a.RF
        pb.P = a.P
        return pb.action_grad(np.asarray(XP, dtype=np.float64), 1.0)[3]
    adolc.function, adolc.gradient = fn, gr
    opts = {'gtol': 1.0e-12, 'ftol': 1.0e-12, 'maxfun': 1000000, 'maxiter': 1000000}   # nnet_twin_anneal.py:133
    X0in = X0.copy()
    nits = []
    import scipy.optimize as so
    real_min = so.minimize

    def spy(*args, **kw):
        r = real_min(*args, **kw)
        nits.append((r.nit, r.nfev, r.status))
        return r
    so.minimize = spy
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            a.anneal(X0, P0.copy(), alpha, betas, RM, RF0, Pidx, Lidx=Lidx, method='L-BFGS-B',
                     opt_args=opts, adolcID=0)
    finally:
        so.minimize = real_min
    rec = dict(structure=structure, M=M, act=act, din=din, dout=dout, X0=X0in, P0=P0, Pidx=np.asarray(Pidx, dtype=int),
               alpha=alpha, beta=np.asarray(betas), RM=RM, RF0=RF0, A_array=a.A_array, me_array=a.me_array,
               fe_array=a.fe_array, minpaths_last=a.minpaths[-1], nit=np.array([n[0] for n in nits]),
               nfev=np.array([n[1] for n in nits]), status=np.array([n[2] for n in nits]))
    rec["_minpaths"] = np.array(a.minpaths)     # the reference's minimiser at every rung (va_nnet.py:510): its own fixture file
    print("%-30s evals=%d A=%s nit=%s" % (name, counts["g"], a.A_array[[0, -1]], rec["nit"]))
    return rec


def main():
    vn = _refload.load_reference("va_nnet")
    if "--only-mnist" in sys.argv:
        flat = {}
        for c, rec in single_cases(vn, mnist_only=True).items():
            for k, v in rec.items():
                flat["%s/%s" % (c, k)] = v
        np.savez_compressed(os.path.join(GOLD, "nnet_mnist.npz"), **flat)
        return
    if "--only-ladder-paths" in sys.argv:
        # The two ladders again, with the reference's minimiser at EVERY rung (va_nnet.py:510), for rung-local parity:
        # a self-contained record (tables + paths of ONE run) in nnet_ladder_paths.npz.  BLAS is held to one thread: with
        # several, two runs of the reference's own twin ladder agree on rungs 0-7 only (rung 8 to 8e-15, then
        # 398 against 576 iterations at rung 9 and actions up to 49 % apart mid-ladder, 5e-6 at the top) -- the
        # long minimisations amplify the last bit of np.dot's summation order.
        from threadpoolctl import threadpool_limits
        old = np.load(os.path.join(GOLD, "nnet.npz"))
        out = {}
        with threadpool_limits(limits=1):
            for name, args, kw in (("g7_twin_ladder", (twin.nnet_structure(20, 10, 10, 10), 2, "sigmoid", 1.1, np.arange(0, 436, 15)), {}),
                                   ("g7_small_tanh_ladder", ([5, 8, 3], 6, "tanh", 1.5, np.arange(0, 40, 2)), {"seed": 1})):
                rec = ladder_case(vn, name, *args, **kw)
                again = ladder_case(vn, name, *args, **kw)
                assert np.array_equal(rec["A_array"], again["A_array"]) and np.array_equal(rec["_minpaths"], again["_minpaths"]), name
                same = int(np.argmin(np.append(old[name + "/A_array"] == rec["A_array"], False)))
                print("%s: rungs 0..%d bit-equal to the run committed in nnet.npz" % (name, same - 1))
                assert abs(old[name + "/A_array"][0] - rec["A_array"][0]) <= 1e-10 * rec["A_array"][0], name      # (rung 0 at least, to rounding)
                for k in ("A_array", "me_array", "fe_array", "nit", "nfev", "status"):
                    out["%s/%s" % (name, k)] = rec[k]
                out[name + "/minpaths"] = rec["_minpaths"]
        np.savez_compressed(os.path.join(GOLD, "nnet_ladder_paths.npz"), **out)
        return
    cases = single_cases(vn)
    # the example's ladder is alpha=1.1, beta=0..435; every 15th rung keeps the run short
    cases["g7_twin_ladder"] = ladder_case(vn, "g7_twin_ladder", twin.nnet_structure(20, 10, 10, 10), 2, "sigmoid",
                                          1.1, np.arange(0, 436, 15))
    cases["g7_small_tanh_ladder"] = ladder_case(vn, "g7_small_tanh_ladder", [5, 8, 3], 6, "tanh", 1.5,
                                                np.arange(0, 40, 2), seed=1)
    flat = {}
    for c, rec in cases.items():
        for k, v in rec.items():
            if not k.startswith("_"):
                flat["%s/%s" % (c, k)] = v
    path = os.path.join(GOLD, "nnet.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, os.path.getsize(path))
    flat = {}
    for c, rec in extra_cases(vn).items():
        for k, v in rec.items():
            flat["%s/%s" % (c, k)] = v
    np.savez_compressed(os.path.join(GOLD, "nnet_extra.npz"), **flat)


if __name__ == "__main__":
    main()
