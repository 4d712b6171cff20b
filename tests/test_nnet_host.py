"""CPU: host side of the va_nnet drop-in (varanneal_amd/va_nnet.py) -- argument handling,
ladder bookkeeping, parameter scatter, the reference's file formats.  The device is replaced
(in this module only) by a stand-in built on the NumPy oracle + SciPy; the real device path
is covered by tests/test_gpu_nnet.py."""
import numpy as np
import pytest

import va_nnet_oracle as vno
from _util import load_npz_cases
from varanneal_amd import _capi, twin, va_nnet

OPTS = {'gtol': 1.0e-12, 'ftol': 1.0e-12, 'maxfun': 1000000, 'maxiter': 1000000}


class OracleNnetProblem(object):
    """Test double for _capi.NnetProblem (same methods/shapes), arithmetic by oracle/."""

    def __init__(self, batch, structure, data_in, data_out, Lidx, RM, RF0, P, Pidx, act="sigmoid", **kw):
        P = np.asarray(P, dtype=np.float64).reshape(batch, -1)
        RMv = np.asarray(RM, dtype=np.float64) if np.ndim(RM) else float(RM)
        self.B = batch
        self.pbs = [vno.NnetProblem(structure, data_in, data_out, Lidx, RMv, RF0, P[b], Pidx, act=act)
                    for b in range(batch)]
        self.NDens = self.pbs[0].NDens

    def close(self):
        pass

    def action_grad(self, XP, rf_scale=1.0, want_grad=True):
        r = [pb.action_grad(XP[b], rf_scale) for b, pb in enumerate(self.pbs)]
        A, me, fe = (np.array([x[i] for x in r]) for i in range(3))
        return A, me, fe, (np.array([x[3] for x in r]) if want_grad else None)

    def _min(self, pb, xp, rf, opt_args):
        import scipy.optimize as opt
        res = opt.minimize(lambda z: pb.action_grad(z, rf)[::3], xp, method='L-BFGS-B', jac=True, options=opt_args)
        A, me, fe = pb.action(res.x, rf)
        return res.x, res.fun, me, fe, res.status, res.nit, res.nfev

    def minimize_lbfgs(self, XP, rf_scale, opt_args=None):
        rows = [self._min(pb, XP[b], rf_scale, opt_args) for b, pb in enumerate(self.pbs)]
        keys = ("x", "A", "me", "fe", "status", "nit", "nfev")
        return {k: np.array([r[i] for r in rows]) for i, k in enumerate(keys)}

    def anneal(self, XP, rf_scale, opt_args=None, want_paths=False, **kw):
        res = []
        for b, pb in enumerate(self.pbs):
            xp = np.array(XP[b]); rows = []
            for rf in rf_scale:
                r = self._min(pb, xp, rf, opt_args)
                xp = r[0]
                rows.append(r)
            res.append(rows)
        g = lambda i, dt=np.float64: np.array([[r[i] for r in rows] for rows in res], dtype=dt)
        return dict(x=None, A=g(1), me=g(2), fe=g(3), status=g(4, np.int32), nit=g(5, np.int32),
                    nfev=g(6, np.int64), minpaths=g(0), pest=None)


@pytest.fixture
def fake_device(monkeypatch):
    monkeypatch.setattr(_capi, "NnetProblem", OracleNnetProblem)


@pytest.fixture(scope="module")
def gold():
    return load_npz_cases("nnet.npz")


def test_activation_recognition():
    assert va_nnet.recognise_activation(twin.sigmoid) == "sigmoid"
    assert va_nnet.recognise_activation(lambda x, W, b: np.tanh(W.dot(x) + b)) == "tanh"
    assert va_nnet.recognise_activation(lambda x, W, b: np.dot(W, x) + b) == "linear"
    assert va_nnet.recognise_activation(lambda x, W, b: np.maximum(np.dot(W, x) + b, 0.0)) == "relu"
    assert va_nnet.recognise_activation(lambda x, W, b: np.log(1.0 + np.exp(np.dot(W, x) + b))) == "softplus"
    assert va_nnet.recognise_activation(lambda x, W, b: np.sin(np.dot(W, x) + b)) is None
    assert va_nnet.recognise_activation("tanh") == "tanh"


def _setup(c, act=twin.sigmoid):
    a = va_nnet.Annealer()
    a.set_structure(c["structure"])
    a.set_activation(act)
    a.set_input_data(c["din"])
    a.set_output_data(c["dout"])
    return a


@pytest.mark.parametrize("fused", [True, False])
def test_twin_flow_follows_reference(fake_device, gold, fused, tmp_path):
    """examples/nnet_twin/nnet_twin_anneal.py:122-146 with the golden run's inputs."""
    c = gold["g7_twin_ladder"]
    nb = 5
    a = _setup(c)
    X0 = c["X0"].copy(); P0 = c["P0"].copy()
    a.anneal(X0, P0, float(c["alpha"]), c["beta"][:nb], float(c["RM"]), float(c["RF0"]), list(c["Pidx"]),
             Lidx=[np.arange(10), np.arange(10)], method='L-BFGS-B', opt_args=OPTS, adolcID=0, verbose=False,
             fused=fused)
    assert a.NDens == 400 and a.NP == 2090 and a.NPest == 1900
    assert list(a.nit_array[:2]) == list(c["nit"][:2])
    assert np.all(np.abs(a.A_array[:2] - c["A_array"][:2]) <= 1e-8 * c["A_array"][:2])
    assert np.all(np.abs(a.A_array[:4] - c["A_array"][:4]) <= 1e-4 * c["A_array"][:4])
    assert a.minpaths.shape == (nb, 400 + 2090)
    # init_to_data wrote the observations into the caller's X0 (va_nnet.py:423-430)
    assert np.array_equal(X0.reshape(2, 200)[:, :10], c["din"])
    # biases were not estimated: they stay at their fixed value in every stored step
    boff = twin.nnet_param_layout(c["structure"])[1]
    assert np.all(a.minpaths[:, 400 + boff[3]:400 + boff[3] + 10] == 0.0)
    assert np.array_equal(a.P, a.minpaths[-1, 400:])
    # savers: shapes of the reference's files
    a.save_io(str(tmp_path / "io.npy")); a.save_Wb(str(tmp_path / "W.npy"), str(tmp_path / "b.npy"))
    a.save_action_errors(str(tmp_path / "aerr.npy")); a.save_states(str(tmp_path / "st.npy"))
    a.save_params(str(tmp_path / "p.npy"))
    assert np.load(str(tmp_path / "io.npy")).shape == (2, nb, 2, 10)
    assert np.load(str(tmp_path / "W.npy")).shape == (nb, 19, 10, 10)
    assert np.load(str(tmp_path / "b.npy")).shape == (nb, 19, 10)
    assert np.load(str(tmp_path / "st.npy")).shape == (2, nb, 20, 10)
    assert np.load(str(tmp_path / "p.npy")).shape == (nb, 2090)
    ae = np.load(str(tmp_path / "aerr.npy"))
    assert ae.shape == (nb, 5) and np.allclose(ae[:, 1], ae[:, 2] + ae[:, 3], rtol=1e-9)
    assert np.allclose(ae[:, 4] * float(c["RF0"]) * float(c["alpha"]) ** ae[:, 0], ae[:, 3])


def test_ragged_savers_and_partial_pidx(fake_device, tmp_path):
    s = [5, 8, 3]; M = 4
    din, dout, _ = twin.make_nnet_twin(s, M)
    X0, P0, Pidx = twin.nnet_initial_guess(s, M, 1, weights_only=True)
    a = va_nnet.Annealer()
    a.set_structure(s); a.set_activation("tanh"); a.set_input_data(din); a.set_output_data(dout)
    a.anneal(X0, P0, 1.5, np.arange(3), [2.0, 3.0], 0.01, Pidx, method='L-BFGS-B', opt_args=OPTS, verbose=False)
    a.save_states(str(tmp_path / "st.npy")); a.save_Wb(str(tmp_path / "W.npy"), str(tmp_path / "b.npy"))
    st = np.load(str(tmp_path / "st.npy"), allow_pickle=True)
    assert st.shape == (M, 3, 3) and st[1, 2, 1].shape == (8,)
    W = np.load(str(tmp_path / "W.npy"), allow_pickle=True)
    assert W.shape == (3, 2) and W[0, 1].shape == (3, 8)
    Wl, bl = a.weights_biases()
    assert np.array_equal(Wl[1], W[2, 1]) and np.all(bl[0] == 0.0)


def test_batched_and_stepwise(fake_device):
    s = [4, 6, 2]; M = 3; B = 2
    din, dout, _ = twin.make_nnet_twin(s, M)
    g = [twin.nnet_initial_guess(s, M, b) for b in range(B)]
    X0 = np.array([x[0] for x in g]); P0 = np.array([x[1] for x in g]); Pidx = g[0][2]
    a = va_nnet.Annealer()
    a.set_structure(s); a.set_activation(twin.sigmoid); a.set_input_data(din); a.set_output_data(dout)
    a.anneal(X0.copy(), P0.copy(), 2.0, np.arange(4), 10.0, 1e-3, Pidx, opt_args=OPTS, verbose=False, fused=False)
    b1 = va_nnet.Annealer()
    b1.set_structure(s); b1.set_activation(twin.sigmoid); b1.set_input_data(din); b1.set_output_data(dout)
    b1.anneal(X0[1].copy(), P0[1].copy(), 2.0, np.arange(4), 10.0, 1e-3, Pidx, opt_args=OPTS, verbose=False)
    assert a.A_array.shape == (B, 4) and a.minpaths.shape == (B, 4, 3 * 12 + 44)
    assert np.allclose(a.A_array[1], b1.A_array, rtol=1e-12) and np.allclose(a.minpaths[1], b1.minpaths)
    with pytest.raises(ValueError):
        a.save_io("x.npy")


def test_bad_inputs_fail_loudly(fake_device):
    s = [4, 6, 2]; M = 3
    din, dout, _ = twin.make_nnet_twin(s, M)
    X0, P0, Pidx = twin.nnet_initial_guess(s, M, 0)
    a = va_nnet.Annealer()
    with pytest.raises(ValueError):
        a.anneal(X0, P0, 2.0, np.arange(2), 1.0, 1e-3, Pidx)
    a.set_structure(s); a.set_input_data(din); a.set_output_data(dout)
    a.set_activation(lambda x, W, b: np.sin(np.dot(W, x)) + b)      # not g(W.x + b): cannot become a kernel epilogue
    with pytest.raises(NotImplementedError):
        a.anneal(X0, P0, 2.0, np.arange(2), 1.0, 1e-3, Pidx)
    a.set_activation(twin.sigmoid)
    with pytest.raises(ValueError):
        a.anneal(X0[:-1], P0, 2.0, np.arange(2), 1.0, 1e-3, Pidx)
    with pytest.raises(ValueError):
        a.anneal(X0, P0[:-1], 2.0, np.arange(2), 1.0, 1e-3, Pidx)
    with pytest.raises(ValueError):
        a.anneal(X0, P0, 2.0, np.arange(2), 1.0, 1e-3, Pidx, method='LM')
    with pytest.raises(ValueError):
        a.anneal(X0, P0, 2.0, np.arange(2), 1.0, 1e-3, Pidx, Lidx=[[0, 1], [0]])     # data has 4 / 2 columns
