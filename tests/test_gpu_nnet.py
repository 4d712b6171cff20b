"""GPU: the feed-forward-network action (csrc/va_nnet.hip, MFMA f64 products) through the
C-ABI against what the reference's va_nnet.Annealer produced (tests/golden/nnet.npz) and,
at sizes the goldens do not reach, against the NumPy oracle."""
import numpy as np
import pytest

import va_nnet_oracle as vno
from _util import load_npz_cases
from varanneal_amd import _capi, twin

pytestmark = pytest.mark.gpu
SINGLE = ["g6_twin_rf1", "g6_twin_rf1e6", "g6_twin_noinit_rf1e4", "g6_ragged_full_rm2", "g6_ragged_tanh",
          "g6_linear_wide", "g6_mnistlike_100_30_10"]


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("nnet.npz")


def _rm(c):
    return c["RM"] if np.ndim(c["RM"]) else float(c["RM"])


@pytest.mark.parametrize("name", SINGLE)
def test_single_eval_matches_reference(gold, name):
    c = gold[name]
    B = 3
    rng = np.random.RandomState(11)
    XP = c["XP"]
    XPb = np.stack([XP, XP + 0.01 * rng.randn(XP.size), XP])
    pr = _capi.NnetProblem(B, c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                           np.tile(c["P"], (B, 1)), c["Pidx"], act=str(c["act"]))
    A, me, fe, g = pr.action_grad(XPb, float(c["rf_scale"]))
    for b in (0, 2):
        assert abs(A[b] - c["A"]) <= 1e-12 * abs(c["A"])
        assert abs(me[b] - c["me"]) <= 1e-12 * abs(c["A"]) and abs(fe[b] - c["fe"]) <= 1e-12 * abs(c["A"])
        assert np.abs(g[b] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()
    assert np.array_equal(g[0], g[2]) and A[1] != A[0]
    # the perturbed seed against the oracle
    pb = vno.NnetProblem(c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                         c["P"], c["Pidx"], act=str(c["act"]))
    A1, me1, fe1, g1 = pb.action_grad(XPb[1], float(c["rf_scale"]))
    assert abs(A[1] - A1) <= 1e-12 * abs(A1) and np.abs(g[1] - g1).max() <= 1e-10 * np.abs(g1).max()
    pr.close()


def test_mnist_shape_single_eval_matches_reference():
    """the tutorial's 784-30-10 network with M = 2 against the reference's own value and complex-step gradient"""
    c = load_npz_cases("nnet_mnist.npz")["g6_mnist_784_30_10"]
    with _capi.NnetProblem(1, c["structure"], c["din"], c["dout"], [c["Lin"], c["Lout"]], _rm(c), float(c["RF0"]),
                           c["P"][None, :], c["Pidx"], act=str(c["act"])) as pr:
        A, me, fe, g = pr.action_grad(c["XP"][None, :], float(c["rf_scale"]))
    assert abs(A[0] - c["A"]) <= 1e-12 * abs(c["A"]) and abs(fe[0] - c["fe"]) <= 1e-12 * abs(c["A"])
    assert np.abs(g[0] - c["grad"]).max() <= 1e-10 * np.abs(c["grad"]).max()


@pytest.mark.parametrize("structure,M,act,weights_only", [
    ([64, 64, 64, 64], 256, "sigmoid", False),      # tiles exactly full
    ([70, 33, 129, 5], 300, "tanh", True),          # ragged in every dimension, 2 example chunks
    ([784, 30, 10], 37, "sigmoid", False),          # tutorial MNIST shape (VarAnneal_tutorial.ipynb:3413-3415)
    ([40, 50, 20], 70, "relu", False),
    ([9, 14, 6], 12, "softplus", False),             # single-kernel path
    ([9, 14, 6], 12, "relu", True),
])
def test_larger_shapes_against_oracle(structure, M, act, weights_only):
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(0, structure[-1], 2)]
    dout = dout[:, Lidx[1]]
    B = 2
    X0, P0, XPs = [], [], []
    for b in range(B):
        x, p, Pidx = twin.nnet_initial_guess(structure, M, b, weights_only)
        p = p + 0.01 * np.random.RandomState(b).randn(p.size)          # non-zero biases
        X0.append(x); P0.append(p); XPs.append(np.append(x, p[Pidx]))
    pr = _capi.NnetProblem(B, structure, din, dout, Lidx, [2.0, 3.0], 0.7, np.array(P0), Pidx, act=act)
    A, me, fe, g = pr.action_grad(np.array(XPs), 1.5)
    for b in range(B):
        pb = vno.NnetProblem(structure, din, dout, Lidx, np.array([2.0, 3.0]), 0.7, P0[b], Pidx, act=act)
        A1, me1, fe1, g1 = pb.action_grad(XPs[b], 1.5)
        assert abs(A[b] - A1) <= 1e-12 * abs(A1) and abs(me[b] - me1) <= 1e-12 * abs(A1)
        assert np.abs(g[b] - g1).max() <= 1e-10 * np.abs(g1).max()
    pr.close()


OPTS = {'gtol': 1.0e-12, 'ftol': 1.0e-12, 'maxfun': 1000000, 'maxiter': 1000000}


def _annealer(c, act):
    from varanneal_amd import va_nnet
    a = va_nnet.Annealer()
    a.set_structure(c["structure"]); a.set_activation(act)
    a.set_input_data(c["din"]); a.set_output_data(c["dout"])
    return a


def test_twin_ladder_on_device(gold):
    """examples/nnet_twin/nnet_twin_anneal.py flow (every 15th rung of its ladder) with the
    L-BFGS loop on the device, against the reference's own anneal() + SciPy."""
    c = gold["g7_twin_ladder"]
    a = _annealer(c, twin.sigmoid)
    a.anneal(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), c["beta"], float(c["RM"]), float(c["RF0"]),
             list(c["Pidx"]), Lidx=[np.arange(10), np.arange(10)], method='L-BFGS-B', opt_args=OPTS, adolcID=0,
             verbose=False)
    # Arbitration: trajectories of long minimisations are chaotic in the
    # last bits -- at rung 0 the device and SciPy agree to 1e-16 after iteration 1, to 1e-11 after
    # iteration 2 (a 10-evaluation line search) and have separated by iteration 3.  So every rung is
    # judged from ITS OWN start point (the device's previous minimiser) against the arbiter: the NumPy
    # oracle of the action under the oracle's own L-BFGS (the optimiser the device restates).
    rows = []
    for k in range(len(c["beta"])):
        xp0 = a._xp0(k); rf = float(a._rf_scale[k])
        pbo = vno.NnetProblem(c["structure"], c["din"], c["dout"], [np.arange(10), np.arange(10)], float(c["RM"]),
                              float(c["RF0"]), a._mp[0, k - 1 if k else 0, a.NDens:], list(c["Pidx"]))
        xo, Ao, sto, nito, nfevo = pbo.minimize_lbfgs(xp0[0], rf, OPTS)
        assert sto == 0, k
        rows.append((k, a.A_array[k], Ao, abs(a.A_array[k] - Ao) / Ao, int(a.nit_array[k]), nito))
    for r in rows:
        print("   rung %2d  device %.6e  arbiter %.6e  rel %.1e  nit %d / %d" % r)
    # Every rung must agree with the arbiter to 1e-3 (measured: <= 2e-6 on 29 of 30 rungs, including
    # minimisations of several hundred iterations).  The exception is recorded, not papered over: rung 8
    # takes ~1000 iterations along a flat valley and stops on ftol = 1e-12 ABSOLUTE (SciPy's
    # max(|f|, 1)) at A ~ 1e-8, i.e. at a relative decrease of 1e-4 per iteration --
    # measured: device 9.009e-09 after 1133 iterations, arbiter 1.296e-08 after 955.  Such a rung must
    # be a long one and the device must not end higher than the arbiter by more than half.
    off = [r for r in rows if r[3] > 1e-3]
    assert len(off) <= 1 and all(min(r[4], r[5]) >= 500 and r[1] <= 1.5 * r[2] for r in off), off
    # against the reference's own anneal() + SciPy (one long trajectory from rung 0): the
    # well-conditioned bottom of the ladder reaches the reference's minima; further up the two
    # trajectories sit in neighbouring minima of comparable depth -- recorded, not asserted tightly
    assert abs(a.A_array[0] - c["A_array"][0]) <= 1e-3 * c["A_array"][0]
    rel = np.abs(a.A_array - c["A_array"]) / c["A_array"]
    print("twin ladder vs reference goldens: rel. deviation per rung", np.array2string(rel, precision=2))
    # (the top of the ladder is pinned rung by rung against the reference itself, from the reference's own start points:
    # test_twin_ladder_every_rung_from_the_references_own_start_point -- 29 of 30 rungs within 1e-3; two runs of the
    # REFERENCE on this machine with different BLAS thread counts agree on rungs 0-7 only, oracle/gen_golden_nnet.py)
    assert np.all(rel[:8] <= 1e-2)
    # What the twin experiment is for (nnet_twin_anneal.py:101-138): the weights that produced the data.  With M = 2
    # examples 1900 weights are not identifiable -- the reference's own run ends 9.18 (rms) away from the true weights
    # having started 0.08 away, the device's 23.9 away (measured) -- so recovered weights cannot arbitrate anything here;
    # they are recorded.  What both runs must have in common is what the data do determine: the final network
    # reproduces the twin's outputs as the reference's does (measurement error), with a model error that stays a small part of A.
    _, _, Ptrue = twin.make_nnet_twin(np.array(c["structure"]), int(c["M"]))
    Pidx = np.asarray(c["Pidx"])
    err_ref = np.sqrt(np.mean((c["minpaths_last"][400:][Pidx] - Ptrue[Pidx]) ** 2))
    err_dev = np.sqrt(np.mean((a.P[Pidx] - Ptrue[Pidx]) ** 2))
    print("recovered weights, rms error against the twin's true weights: device %.4f, reference %.4f; final me %.3e / %.3e, fe %.3e / %.3e"
          % (err_dev, err_ref, a.me_array[-1], c["me_array"][-1], a.fe_array[-1], c["fe_array"][-1]))
    assert np.isfinite(err_dev)
    # (measured: me 0.40011 / 0.40023, fe 1.08e-3 / 7.3e-6 -- the data term is the reference's to 3e-4; the model term,
    # 0.3 % of A up here, is where the two neighbouring minima differ)
    assert abs(a.me_array[-1] - c["me_array"][-1]) <= 1e-2 * c["me_array"][-1] and a.fe_array[-1] <= 1e-2 * a.A_array[-1]
    assert np.all(a.exitflags == 0)
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    A, g = a.A_gradA_taped(np.append(a.minpaths[-1][:400], a.P[c["Pidx"]]))
    assert abs(A - a.A_array[-1]) <= 1e-12 * A
    a.close()


def test_twin_ladder_every_rung_from_the_references_own_start_point(gold):
    """Rung-local parity against the REFERENCE (tests/golden/nnet_ladder_paths.npz: tables and per-rung minimisers of one
    single-BLAS-thread run of its own anneal() + SciPy, oracle/gen_golden_nnet.py --only-ladder-paths).  Rung k starts
    from the reference's minimiser of rung k-1 (va_nnet.py:455-472) and is compared with the reference's rung k:
    (A, me, fe) within 1e-3.  Rungs that end elsewhere are listed with both minima."""
    c = gold["g7_twin_ladder"]
    ref = load_npz_cases("nnet_ladder_paths.npz")["g7_twin_ladder"]
    s, Pidx = c["structure"], np.asarray(c["Pidx"])
    NDens = int(c["M"]) * int(np.sum(s))
    nb = len(c["beta"])
    assert ref["minpaths"].shape == (nb, NDens + len(c["P0"]))
    rf = float(c["alpha"]) ** np.asarray(c["beta"], dtype=float)
    Lidx = [np.arange(s[0]), np.arange(s[-1])]
    rows = []
    for k in range(nb):
        if k:
            X, P = ref["minpaths"][k - 1, :NDens], ref["minpaths"][k - 1, NDens:]
        else:
            # (init_to_data, va_nnet.py:425-430: observed input / output neurons start at the data)
            a0 = _annealer(c, twin.sigmoid)
            a0.anneal_init(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), c["beta"][:1], float(c["RM"]), float(c["RF0"]),
                           list(Pidx), Lidx=Lidx, method='L-BFGS-B', opt_args=OPTS, adolcID=0, verbose=False)
            X, P = a0._xp0(0)[0][:NDens], c["P0"]
            a0.close()
        with _capi.NnetProblem(1, s, c["din"], c["dout"], Lidx, float(c["RM"]), float(c["RF0"]), P[None, :], Pidx) as pb:
            r = pb.minimize_lbfgs(np.append(X, P[Pidx])[None, :], rf[k], OPTS)
        dev = np.array([r["A"][0], r["me"][0], r["fe"][0]])
        want = np.array([ref["A_array"][k], ref["me_array"][k], ref["fe_array"][k]])
        rows.append((k, np.abs(dev - want) / abs(want[0]), dev, want, int(r["nit"][0]), int(ref["nit"][k]), int(r["status"][0]), int(ref["status"][k])))
    off = [q for q in rows if q[1].max() > 1e-3]
    for q in off:
        print("   twin rung %2d: device A %.6e me %.3e fe %.3e (nit %d, status %d) | reference A %.6e me %.3e fe %.3e (nit %d, status %d)"
              % (q[0], q[2][0], q[2][1], q[2][2], q[4], q[6], q[3][0], q[3][1], q[3][2], q[5], q[7]))
    print("   statuses other than 0: device %s, reference %s" % ([(q[0], q[6]) for q in rows if q[6]], [(q[0], q[7]) for q in rows if q[7]]))
    print("twin ladder: %d of %d rungs within 1e-3 of the reference's (A, me, fe) from the reference's start; %d with its iteration count"
          % (nb - len(off), nb, sum(1 for q in rows if q[4] == q[5])))
    assert nb - len(off) >= 27, [q[0] for q in off]
    assert rows[-1][1].max() <= 1e-3


def test_stepwise_equals_fused_and_batch_independent(gold):
    c = gold["g7_small_tanh_ladder"]
    s, M = c["structure"], int(c["M"])
    B, nb = 3, 6
    g = [twin.nnet_initial_guess(s, M, b) for b in range(B)]
    X0 = np.array([x[0] for x in g]); P0 = np.array([x[1] for x in g]); Pidx = g[0][2]
    act = lambda x, W, b: np.tanh(np.dot(W, x) + b)
    args = (float(c["alpha"]), c["beta"][:nb], float(c["RM"]), float(c["RF0"]), Pidx)
    f = _annealer(c, act); f.anneal(X0.copy(), P0.copy(), *args, opt_args=OPTS, verbose=False)
    st = _annealer(c, act); st.anneal(X0.copy(), P0.copy(), *args, opt_args=OPTS, verbose=False, fused=False)
    one = _annealer(c, act); one.anneal(X0[1].copy(), P0[1].copy(), *args, opt_args=OPTS, verbose=False)
    assert np.array_equal(f.A_array, st.A_array) and np.array_equal(f.minpaths, st.minpaths)
    assert np.array_equal(f.A_array[1], one.A_array) and np.array_equal(f.minpaths[1], one.minpaths)
    # seed 1 of the golden run is this seed.  Rung 0 reproduces the reference; rung 1 takes ~200
    # iterations and its end point depends on summation order (the two device paths agree on (nit, nfev)
    # for 40 iterations and then part ways), so it is arbitrated: from rung 1's own start point the
    # device and the NumPy oracle under the oracle's L-BFGS must find the same minimum
    assert abs(f.A_array[1, 0] - c["A_array"][0]) <= 1e-3 * c["A_array"][0]
    pbo = vno.NnetProblem(s, c["din"], c["dout"], [np.arange(s[0]), np.arange(s[-1])], float(c["RM"]), float(c["RF0"]),
                          f._mp[1, 0, f.NDens:], Pidx, act="tanh")
    xo, Ao, sto, nito, nfevo = pbo.minimize_lbfgs(f._xp0(1)[1], float(f._rf_scale[1]), OPTS)
    print("small tanh ladder rung 1: device %.6e arbiter %.6e reference %.6e" % (f.A_array[1, 1], Ao, c["A_array"][1]))
    # Recorded (rung 1, ~200 iterations along a flat valley; every run below stops by the same ftol rule):
    #   reference's own run              2.018360e-06
    #   device, round-2 builds           2.018944e-06, 2.022566e-06   (two builds that differ in FMA contraction only)
    #   arbiter from the device's start  2.023559e-06, 3.027893e-06   (starts that differ in the last bits)
    # i.e. last-bit changes of the start or of the summation order move the stopping point by up to 2.6e-3
    # among the runs that reach the valley floor, and the arbiter itself once stopped 50 % higher.  What is
    # asserted: the device stops on the valley floor -- within 5e-3 of the reference's end point -- and never
    # above the worse of the two CPU end points.
    dev = f.A_array[1, 1]
    assert sto == 0, sto
    assert abs(dev - c["A_array"][1]) <= 5e-3 * c["A_array"][1], (dev, Ao, c["A_array"][1])
    assert dev <= max(Ao, c["A_array"][1]) * (1.0 + 1e-3), (dev, Ao, c["A_array"][1])
    for x in (f, st, one):
        x.close()


def test_bounds_route_uses_device_evaluator(gold):
    c = gold["g7_small_tanh_ladder"]
    s, M = c["structure"], int(c["M"])
    X0, P0, Pidx = twin.nnet_initial_guess(s, M, 1)
    a = _annealer(c, "tanh")
    bounds = [(-3.0, 3.0)] * (M * int(np.sum(s))) + [(-0.05, 0.05)] * len(Pidx)
    a.anneal(X0, P0, float(c["alpha"]), c["beta"][:4], float(c["RM"]), float(c["RF0"]), Pidx, bounds=bounds,
             opt_args=OPTS, verbose=False)
    est = a.minpaths[:, a.NDens:][:, Pidx]
    assert np.all(np.abs(est) <= 0.05 + 1e-15) and np.all(a.exitflags == 0)
    assert np.allclose(a.A_array, a.me_array + a.fe_array, rtol=1e-12)
    a.close()


@pytest.mark.parametrize("structure,M,act", [([1, 1], 1, "sigmoid"), ([3, 2], 1, "tanh"), ([2, 5, 1], 2, "linear"),
                                             ([33, 2], 3, "sigmoid"), ([2, 33], 33, "tanh")])
def test_smallest_networks(structure, M, act):
    """two layers (no hidden layer), one example, single neurons, and shapes just past the
    single-kernel limit (33 > 32) so both evaluators see their edge."""
    rng = np.random.RandomState(1)
    din, dout = rng.randn(M, structure[0]), rng.rand(M, structure[-1])     # (the twin recipe standardises
    Lidx = [np.arange(structure[0]), np.arange(structure[-1])]             # the input: undefined for 1 neuron)
    NP = twin.nnet_param_layout(structure)[2]
    X0, P0, Pidx = rng.rand(M * int(np.sum(structure))), 0.3 * rng.randn(NP), list(range(NP))
    XP = np.append(X0, P0[Pidx])
    pr = _capi.NnetProblem(1, structure, din, dout, Lidx, 2.0, 0.3, P0[None, :], Pidx, act=act)
    A, me, fe, g = pr.action_grad(XP[None, :], 4.0)
    pb = vno.NnetProblem(structure, din, dout, Lidx, 2.0, 0.3, P0, Pidx, act=act)
    A1, me1, fe1, g1 = pb.action_grad(XP, 4.0)
    assert abs(A[0] - A1) <= 1e-12 * abs(A1) and abs(me[0] - me1) <= 1e-12 * abs(A1)
    assert np.abs(g[0] - g1).max() <= 1e-10 * np.abs(g1).max()
    r = pr.minimize_lbfgs(XP[None, :], 4.0, {'gtol': 1e-10, 'ftol': 1e-12, 'maxfun': 500, 'maxiter': 500})
    assert r["A"][0] < A[0] and r["status"][0] in (0, 1)
    pr.close()


@pytest.mark.parametrize("structure,M,act", [([64, 64, 64, 64], 256, "sigmoid"), ([40, 50, 20], 70, "relu"),
                                             ([128, 100, 128, 7], 130, "tanh")])
def test_fused_forward_backward_kernel_matches_oracle(structure, M, act):
    """k_nnet_fb (forward and state-gradient products of every transition in one kernel; tune nnet_fused=1) against the
    NumPy oracle and, bit for bit in A up to rounding, against the separate kernels"""
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(0, structure[-1], 2)]
    dout = dout[:, Lidx[1]]
    X, P, Pidx = twin.nnet_initial_guess(structure, M, 3)
    XP = np.append(X, P[Pidx])[None, :]
    with _capi.NnetProblem(1, structure, din, dout, Lidx, 3.0, 0.02, P[None, :], Pidx, act=act) as pr:
        pr.tune(nnet_fused=0)
        A0, me0, fe0, g0 = pr.action_grad(XP, 7.0)
        pr.tune(nnet_fused=1)
        A1, me1, fe1, g1 = pr.action_grad(XP, 7.0)
        r = pr.minimize_lbfgs(XP, 7.0, {'gtol': 1e-10, 'ftol': 1e-10, 'maxfun': 60, 'maxiter': 40})
    Ao, meo, feo, go = vno.NnetProblem(structure, din, dout, Lidx, 3.0, 0.02, P, Pidx, act=act).action_grad(XP[0], 7.0)
    assert abs(A1[0] - Ao) <= 1e-12 * abs(Ao) and abs(me1[0] - meo) <= 1e-12 * abs(Ao)
    assert np.abs(g1[0] - go).max() <= 1e-10 * np.abs(go).max()
    assert abs(A1[0] - A0[0]) <= 1e-13 * abs(A0[0]) and np.abs(g1[0] - g0[0]).max() <= 1e-12 * np.abs(g0[0]).max()
    assert r["A"][0] < A1[0]



@pytest.mark.parametrize("structure,M,act", [([64, 64, 64], 2048, "sigmoid"), ([70, 100, 33], 2030, "tanh")])
def test_fused_kernel_is_the_default_when_its_blocks_fill_the_chip(structure, M, act):
    """8 seeds x 64 blocks of 32 examples = 512 workgroups: the handle evaluates with k_nnet_fb unasked.  Action and
    gradient against the NumPy oracle and the separate kernels; a minimisation (line-search evaluations: the direction's
    entries are prefetched too) against the separate kernels' iterates"""
    B = 8
    din, dout, _ = twin.make_nnet_twin(structure, M)
    Lidx = [np.arange(structure[0]), np.arange(structure[-1])]
    g = [twin.nnet_initial_guess(structure, M, b) for b in range(B)]
    Pidx = g[0][2]
    P = np.array([x[1] for x in g])
    XP = np.array([np.append(x[0], x[1][Pidx]) for x in g])
    opts = {'gtol': 1e-10, 'ftol': 1e-10, 'maxfun': 40, 'maxiter': 12}
    with _capi.NnetProblem(B, structure, din, dout, Lidx, 3.0, 0.02, P, Pidx, act=act) as pr:
        A1, me1, fe1, g1 = pr.action_grad(XP, 7.0)
        r1 = pr.minimize_lbfgs(XP, 7.0, opts)
        pr.tune(nnet_fused=0)
        A0, me0, fe0, g0 = pr.action_grad(XP, 7.0)
        r0 = pr.minimize_lbfgs(XP, 7.0, opts)
        pr.tune(nnet_fused=1)
        A2, _, _, g2 = pr.action_grad(XP, 7.0)
    assert np.array_equal(A1, A2) and np.array_equal(g1, g2)                  # the default WAS the fused kernel
    assert not np.array_equal(g1, g0)                                          # (and the other path is another summation order)
    assert np.all(np.abs(A1 - A0) <= 1e-13 * np.abs(A0)) and np.abs(g1 - g0).max() <= 1e-12 * np.abs(g0).max()
    for b in (0, B - 1):
        Ao, meo, feo, go = vno.NnetProblem(structure, din, dout, Lidx, 3.0, 0.02, P[b], Pidx, act=act).action_grad(XP[b], 7.0)
        assert abs(A1[b] - Ao) <= 1e-12 * abs(Ao) and abs(me1[b] - meo) <= 1e-12 * abs(Ao)
        assert np.abs(g1[b] - go).max() <= 1e-10 * np.abs(go).max()
    assert np.array_equal(r1["nit"], r0["nit"]) and np.array_equal(r1["nfev"], r0["nfev"])
    assert np.all(np.abs(r1["A"] - r0["A"]) <= 1e-9 * np.abs(r0["A"])) and np.all(r1["A"] < A1)
