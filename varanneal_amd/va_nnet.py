"""Drop-in for the reference's `varanneal.va_nnet.Annealer` (varanneal/va_nnet.py) on MI355X.

Same API as upstream:
    set_structure(structure) / set_activation(f) / set_input_data(data) / set_output_data(data)
    anneal(X0, P0, alpha, beta_array, RM, RF0, Pidx, Lidx=None, init_to_data=True,
           action='A_gaussian', disc='forwardmap', method='L-BFGS-B', bounds=None,
           opt_args=None, adolcID=0)
    anneal_init(...) / anneal_step()
    minpaths, A_array, me_array, fe_array, exitflags, P
    save_states / save_io / save_params / save_Wb / save_action_errors

What runs where: the action, its gradient (three float64 MFMA products per layer, no AD
tape) and the L-BFGS loop over the RF ladder run on the device behind the C-ABI
(include/varanneal_amd.h: va_nnet_problem_create + the shared S1/S2/S3 entry points).
`bounds` / `NCG` / `TNC` run SciPy on the host around the device evaluator.

Extensions (keyword-only, defaults keep upstream behaviour): X0 of shape (B, M*NDnet) and
P0 of shape (B, NP) anneal B initial guesses as one batch; `device`, `verbose`, `fused`.

The activation is the reference's callable f(x, W, b) (examples/nnet_twin/
nnet_twin_anneal.py:20-22); it is matched numerically against the built-in registry
(sigmoid, tanh, linear, relu, softplus of W.x + b) -- anything else raises NotImplementedError.

Upstream defects fixed rather than reproduced: the default `Lidx` is an integer range
(va_nnet.py:326-327 builds a float linspace NumPy refuses as an index); `save_params` slices
the estimated parameters at NDens (va_nnet.py:596 uses NDnet); `save_states` walks the layers
with structure[n+1] (va_nnet.py:534 uses structure[n], equal only for uniform nets);
`exitflags` are written.  Kept: `init_to_data` overwrites the caller's X0 (va_nnet.py:423-430);
`anneal()` always re-initialises (va_nnet.py:276, :450 typo).
"""
import time

import numpy as np

from . import _capi
from ._hipmin import HIPmin, alpha_pow as _alpha_pow

ACT_IMPL = {
    "sigmoid": lambda x, W, b: 1.0 / (1.0 + np.exp(-(np.dot(W, x) + b))),
    "tanh": lambda x, W, b: np.tanh(np.dot(W, x) + b),
    "linear": lambda x, W, b: np.dot(W, x) + b,
    "relu": lambda x, W, b: np.maximum(np.dot(W, x) + b, 0.0),
    "softplus": lambda x, W, b: np.logaddexp(0.0, np.dot(W, x) + b),
}


def recognise_activation(f):
    """Name of the built-in activation `f(x, W, b)` agrees with on random probes, or None."""
    if isinstance(f, str):
        return f if f in ACT_IMPL else None
    tag = getattr(f, "va_act", None)
    if tag in ACT_IMPL:
        return tag
    rng = np.random.RandomState(20260102)
    for name, impl in ACT_IMPL.items():
        ok = True
        for _ in range(3):
            x = rng.randn(5); W = rng.randn(4, 5); b = rng.randn(4)
            try:
                got = np.asarray(f(x, W, b), dtype=np.float64)
            except Exception:
                ok = False
                break
            if got.shape != (4,) or not np.allclose(got, impl(x, W, b), rtol=1e-12, atol=1e-12):
                ok = False
                break
        if ok:
            return name
    return None


class Annealer(HIPmin):
    def __init__(self):
        self.taped = False
        self.annealing_initialized = False
        self.M = 0
        self.structure = None
        self.f = None
        self._act = None
        self._pb = None

    # ------------------------------------------------------------------ setup (va_nnet.py:59-106)
    def set_structure(self, structure):
        self.structure = np.asarray(structure, dtype=int)
        self.N = len(self.structure)
        if self.M > 0:
            self.NDnet = int(np.sum(self.structure))
            self.NDens = self.NDnet * self.M

    def set_activation(self, f):
        self.f = f
        self._act = recognise_activation(f)

    def _set_data(self, data):
        data = np.asarray(data, dtype=np.float64)
        if data.ndim == 1:
            data = np.array([data])
        if self.M == 0:
            self.M = data.shape[0]
        if self.structure is not None:
            self.NDnet = int(np.sum(self.structure))
            self.NDens = self.NDnet * self.M
        return data

    def set_input_data(self, data):
        self.data_in = self._set_data(data)

    def set_output_data(self, data):
        self.data_out = self._set_data(data)

    # ------------------------------------------------------------------ annealing
    def anneal(self, X0, P0, alpha, beta_array, RM, RF0, Pidx, Lidx=None,
               init_to_data=True, action='A_gaussian', disc='forwardmap',
               method='L-BFGS-B', bounds=None, opt_args=None, adolcID=0,
               *, device=0, verbose=True, fused=None):
        """Full ladder (va_nnet.py:267-286)."""
        if self.annealing_initialized is False:
            self.anneal_init(X0, P0, alpha, beta_array, RM, RF0, Pidx, Lidx, init_to_data, action, disc,
                             method, bounds, opt_args, adolcID, device=device, verbose=verbose)
        if fused is None:
            fused = self._device_minimiser
        if fused:
            if not self._device_minimiser:
                raise ValueError("fused=True needs method='L-BFGS-B' with bounds=None")
            self._anneal_fused()
            return
        for _ in self.beta_array:
            if self.verbose:
                print('------------------------------')
                print('Step %d of %d' % (self.betaidx + 1, len(self.beta_array)))
                print('beta = %d, RF = %.8e' % (self.beta, self.RF))
                print('')
            self.anneal_step()

    def anneal_init(self, X0, P0, alpha, beta_array, RM, RF0, Pidx, Lidx=None,
                    init_to_data=True, action='A_gaussian', disc='forwardmap',
                    method='L-BFGS-B', bounds=None, opt_args=None, adolcID=0,
                    *, device=0, verbose=True):
        """va_nnet.py:288-450."""
        if self.structure is None or self.M == 0:
            raise ValueError("set_structure / set_input_data / set_output_data must be called first")
        if self.f is None:
            raise ValueError("set_activation must be called first")
        if self._act is None:
            # any other layer map g(W.x + b) with an elementwise g: trace g, differentiate it, compile the two
            # kernels that apply it (csrc/va_user_act.hip) -- the reference takes any callable (va_nnet.py:71)
            from . import codegen
            try:
                self._act_module = codegen.activation_module_for(self.f)
            except TypeError as e:
                raise NotImplementedError("activation %r: %s (built-in forms: %s)" % (self.f, e, sorted(ACT_IMPL)))
            self._act = _capi.load_act_module(self._act_module["so"])
        if method not in ('L-BFGS-B', 'NCG', 'TNC'):
            raise ValueError("Optimization routine %r not recognized (upstream also lists 'LM', which it "
                             "never implemented for this class)." % (method,))
        if action != 'A_gaussian':
            raise NotImplementedError("only action='A_gaussian' is implemented")
        if disc != 'forwardmap':
            raise ValueError("va_nnet has only disc='forwardmap' (va_nnet.py:260-264)")
        self.method = method
        self.verbose = verbose
        self.opt_args = opt_args
        self.bounds = None if bounds is None else np.array(bounds)

        X0 = np.asarray(X0) if not isinstance(X0, np.ndarray) else X0
        P0 = np.asarray(P0, dtype=np.float64) if not isinstance(P0, np.ndarray) else P0
        if X0.dtype != np.float64:
            raise ValueError("X0 must be a float64 array")
        self._batched = X0.ndim == 2
        if self._batched != (P0.ndim == 2):
            raise ValueError("X0 and P0 must both be single (1-D) or both batched (2-D)")
        self.B = X0.shape[0] if self._batched else 1
        Xf = X0 if self._batched else X0[None, :]                      # views: init_to_data writes through
        Pf = P0 if self._batched else P0[None, :]
        if Xf.shape[1] != self.NDens:
            raise ValueError("X0 must hold M*NDnet = %d states, got %d" % (self.NDens, Xf.shape[1]))
        s = self.structure
        NPnet = int(np.sum(s[1:] * s[:-1] + s[1:]))
        if Pf.shape[1] != NPnet:
            raise ValueError("P0 must hold %d weights and biases for this structure, got %d" % (NPnet, Pf.shape[1]))
        self.P = P0
        self.NP = NPnet
        self.Pidx = list(Pidx)
        self.NPest = len(self.Pidx)

        if Lidx is None:
            self.Lidx = [np.arange(s[0]), np.arange(s[-1])]
        else:
            self.Lidx = [np.asarray(Lidx[0], dtype=int), np.asarray(Lidx[1], dtype=int)]
        self.L = [len(self.Lidx[0]), len(self.Lidx[1])]
        self.Ltot = self.L[0] + self.L[1]
        if self.data_in.shape != (self.M, self.L[0]) or self.data_out.shape != (self.M, self.L[1]):
            raise ValueError("input/output data must have shapes (M, L_in) = (%d, %d) and (M, L_out) = (%d, %d)"
                             % (self.M, self.L[0], self.M, self.L[1]))
        self.RM = RM

        self.alpha = alpha
        self.beta_array = np.asarray(beta_array)
        self.betaidx = 0
        self.beta = self.beta_array[0]
        self.Nbeta = len(self.beta_array)
        if RF0 is not None:
            self.RF0 = RF0
        self.RF = self.RF0 * _alpha_pow(self.alpha, self.beta)
        self._rf_scale = _alpha_pow(self.alpha, self.beta_array)

        if init_to_data:                                               # va_nnet.py:423-430
            Xv = Xf.reshape(self.B, self.M, self.NDnet)
            Xv[:, :, self.Lidx[0]] = self.data_in
            Xv[:, :, self.NDnet - s[-1] + self.Lidx[1]] = self.data_out

        self._mp = np.zeros((self.B, self.Nbeta, self.NDens + self.NP))
        self._mp[:, 0, :self.NDens] = Xf
        self._mp[:, 0, self.NDens:] = Pf
        self._A = np.zeros((self.B, self.Nbeta)); self._me = np.zeros((self.B, self.Nbeta))
        self._fe = np.zeros((self.B, self.Nbeta))
        self._flags = np.zeros((self.B, self.Nbeta), dtype=np.int8)
        self._nit = np.zeros((self.B, self.Nbeta), dtype=np.int32)
        self._nfev = np.zeros((self.B, self.Nbeta), dtype=np.int64)
        self._Pfull = np.array(Pf, dtype=np.float64)
        self.adolcID = adolcID                                         # accepted, unused: there is no tape
        self._device_minimiser = (method == 'L-BFGS-B' and bounds is None)
        if not self._device_minimiser and self.B != 1:
            raise ValueError("bounds / NCG / TNC run SciPy on the host around the device evaluator: one seed only")

        if self._pb is not None:
            self._pb.close()
        self._pb = _capi.NnetProblem(self.B, s, self.data_in, self.data_out, self.Lidx, RM, float(self.RF0),
                                     self._Pfull, self.Pidx, act=self._act,
                                     lbfgs_m=int((opt_args or {}).get("maxcor", 10)), max_beta=self.Nbeta,
                                     keep_paths=1, device=device)
        self.initalized = True                                         # sic (va_nnet.py:450)

    def _view(self, a):
        return a if self._batched else a[0]

    minpaths = property(lambda self: self._view(self._mp))
    A_array = property(lambda self: self._view(self._A))
    me_array = property(lambda self: self._view(self._me))
    fe_array = property(lambda self: self._view(self._fe))
    exitflags = property(lambda self: self._view(self._flags))
    nit_array = property(lambda self: self._view(self._nit))
    nfev_array = property(lambda self: self._view(self._nfev))

    def _xp0(self, k):
        """va_nnet.py:460-473"""
        src = self._mp[:, k - 1 if k > 0 else 0]
        return np.concatenate([src[:, :self.NDens], src[:, self.NDens:][:, self.Pidx]], axis=1)

    def _store(self, k, x, A, me, fe, flag, nit, nfev):
        self._Pfull[:, self.Pidx] = x[:, self.NDens:]                  # va_nnet.py:493-499
        if self._batched:
            self.P[:, self.Pidx] = x[:, self.NDens:]
        else:
            self.P[self.Pidx] = x[0, self.NDens:]
        self._A[:, k] = A; self._me[:, k] = me; self._fe[:, k] = fe      # :502-504
        self._mp[:, k, :self.NDens] = x[:, :self.NDens]; self._mp[:, k, self.NDens:] = self._Pfull   # :505
        self._flags[:, k] = flag; self._nit[:, k] = nit; self._nfev[:, k] = nfev

    def anneal_step(self):
        """One ladder step for every seed (va_nnet.py:452-523)."""
        k = self.betaidx
        XP0 = self._xp0(k)
        rf = float(self._rf_scale[k])
        t0 = time.time()
        if self._device_minimiser:
            r = self._pb.minimize_lbfgs(XP0, rf, self.opt_args)
            x, A, me, fe, flag, nit, nfev = r["x"], r["A"], r["me"], r["fe"], r["status"], r["nit"], r["nfev"]
        else:
            x, A, me, fe, flag, nit, nfev = self._minimize_scipy(XP0, rf)
        self._store(k, x, A, me, fe, flag, nit, nfev)
        if self.verbose:
            print("Optimization complete!")
            print("Time = {0} s".format(time.time() - t0))
            print("Exit flag = {0}".format(flag[0] if self.B == 1 else flag))
            print("Iterations = {0}".format(nit[0] if self.B == 1 else nit))
            print("Obj. function value = {0}\n".format(A[0] if self.B == 1 else A))
        if self.betaidx < len(self.beta_array) - 1:                    # va_nnet.py:508-511
            self.betaidx += 1
            self.beta = self.beta_array[self.betaidx]
            self.RF = self.RF0 * _alpha_pow(self.alpha, self.beta)
        self.taped = False

    def _anneal_fused(self):
        """Remaining ladder steps in one va_anneal call; seeds advance independently."""
        k0 = self.betaidx
        t0 = time.time()
        r = self._pb.anneal(self._xp0(k0), self._rf_scale[k0:], self.opt_args, want_paths=True)
        nb = self.Nbeta - k0
        self._A[:, k0:] = r["A"]; self._me[:, k0:] = r["me"]; self._fe[:, k0:] = r["fe"]
        self._flags[:, k0:] = r["status"]; self._nit[:, k0:] = r["nit"]; self._nfev[:, k0:] = r["nfev"]
        mp = r["minpaths"]                                             # [B][nb][NDens + NPest]
        self._mp[:, k0:, :self.NDens] = mp[:, :, :self.NDens]
        self._mp[:, k0:, self.NDens:] = self._Pfull[:, None, :]
        self._mp[:, k0:, [self.NDens + j for j in self.Pidx]] = mp[:, :, self.NDens:]
        self._Pfull[:] = self._mp[:, -1, self.NDens:]
        if self._batched:
            self.P[:, self.Pidx] = self._Pfull[:, self.Pidx]
        else:
            self.P[self.Pidx] = self._Pfull[0, self.Pidx]
        self.betaidx = self.Nbeta - 1
        self.beta = self.beta_array[self.betaidx]
        self.RF = self.RF0 * _alpha_pow(self.alpha, self.beta)
        if self.verbose:
            print("Ladder of %d steps x %d seed(s): %.3f s, %d action+gradient evaluations"
                  % (nb, self.B, time.time() - t0, int(self._nfev[:, k0:].sum())))

    def _minimize_scipy(self, XP0, rf):
        """bounds / NCG / TNC: SciPy exactly as _autodiffmin.py:72-146 calls it, the device
        kernels in the role of A_gradA_taped."""
        import scipy.optimize as opt
        meth = {'L-BFGS-B': 'L-BFGS-B', 'NCG': 'CG', 'TNC': 'TNC'}[self.method]

        def fg(z):
            A, me, fe, g = self._pb.action_grad(z[None, :], rf)
            return A[0], g[0]
        kw = dict(method=meth, jac=True, options=self.opt_args)
        if meth != 'CG':
            kw["bounds"] = self.bounds
        res = opt.minimize(fg, XP0[0], **kw)
        A, me, fe, _ = self._pb.action_grad(res.x[None, :], rf, want_grad=False)
        return (res.x[None, :], np.array([res.fun]), me, fe, np.array([res.status]), np.array([res.nit]),
                np.array([res.nfev]))

    # ------------------------------------------------------------------ S1 evaluator
    def _eval(self, XP, want_grad):
        XP = np.asarray(XP, dtype=np.float64)
        single = XP.ndim == 1
        X2 = np.tile(XP, (self.B, 1)) if single else XP
        A, me, fe, g = self._pb.action_grad(X2, float(self._rf_scale[self.betaidx]), want_grad=want_grad)
        if single:
            return A[0], me[0], fe[0], (g[0] if want_grad else None)
        return A, me, fe, g

    def A_gaussian(self, XP):
        return self._eval(XP, False)[0]

    A = A_gaussian

    def me_gaussian(self, XP):
        return self._eval(XP, False)[1]

    def fe_gaussian(self, XP):
        return self._eval(XP, False)[2]

    def close(self):
        if self._pb is not None:
            self._pb.close()
            self._pb = None

    # ------------------------------------------------------------------ savers (va_nnet.py:528-650)
    def _layers(self, row):
        """split one example's NDnet states into per-layer arrays"""
        off = np.concatenate([[0], np.cumsum(self.structure)])
        return [row[off[n]:off[n + 1]] for n in range(self.N)]

    def _pack(self, nested, dtype):
        """np.array(nested) when the net is uniform, an object array otherwise (what
        np.array did for ragged lists when the reference was written)."""
        try:
            return np.array(nested, dtype=dtype)
        except ValueError:
            out = np.empty((len(nested), len(nested[0]), len(nested[0][0])), dtype=object)
            for a, xa in enumerate(nested):
                for b, xb in enumerate(xa):
                    for c, xc in enumerate(xb):
                        out[a, b, c] = np.asarray(xc, dtype=dtype)
            return out

    def _one(self):
        if self._batched:
            raise ValueError("savers write one seed's results; index the arrays (minpaths[b], ...) for a batch")
        return self._mp[0]

    def save_states(self, filename, dtype=np.float64, fmt="%.8e"):
        """(M, Nbeta, N layers) neuron states."""
        mp = self._one()
        nested = [[self._layers(mp[b, m * self.NDnet:(m + 1) * self.NDnet]) for b in range(self.Nbeta)]
                  for m in range(self.M)]
        arr = self._pack(nested, dtype)
        if filename.endswith('.npy'):
            np.save(filename, arr, allow_pickle=True)
        else:
            np.savetxt(filename, arr.reshape(-1, arr.shape[-1]), fmt=fmt)

    def save_io(self, filename, dtype=np.float64, fmt="%.8e"):
        """(M, Nbeta, 2): input- and output-layer states (va_nnet.py:553-577)."""
        mp = self._one()
        s = self.structure
        nested = [[[mp[b, m * self.NDnet:m * self.NDnet + s[0]],
                    mp[b, (m + 1) * self.NDnet - s[-1]:(m + 1) * self.NDnet]] for b in range(self.Nbeta)]
                  for m in range(self.M)]
        arr = self._pack(nested, dtype)
        if filename.endswith('.npy'):
            np.save(filename, arr, allow_pickle=True)
        else:
            np.savetxt(filename, arr.reshape(-1, arr.shape[-1]), fmt=fmt)

    def save_params(self, filename, dtype=np.float64, fmt="%.8e"):
        if self.NPest == 0:
            print("WARNING: You did not estimate any parameters.  Writing fixed parameter values to file anyway.")
        savearray = np.array(self._one()[:, self.NDens:])
        if filename.endswith('.npy'):
            np.save(filename, savearray.astype(dtype))
        else:
            np.savetxt(filename, savearray, fmt=fmt)

    def weights_biases(self, beta_idx=-1, seed=0):
        """([W_0, ...], [b_0, ...]) of one ladder step (layout va_nnet.py:194-207)."""
        p = self._mp[seed, beta_idx, self.NDens:]
        s = self.structure
        W, b, o = [], [], 0
        for n in range(self.N - 1):
            W.append(p[o:o + s[n + 1] * s[n]].reshape(s[n + 1], s[n])); o += s[n + 1] * s[n]
            b.append(p[o:o + s[n + 1]]); o += s[n + 1]
        return W, b

    def save_Wb(self, W_filename, b_filename, dtype=np.float64):
        self._one()
        Ws, bs = [], []
        for i in range(self.Nbeta):
            W, b = self.weights_biases(i)
            Ws.append([w.astype(dtype) for w in W]); bs.append([v.astype(dtype) for v in b])

        def pack(nested):
            try:
                return np.array(nested, dtype=dtype)
            except ValueError:
                out = np.empty((len(nested), len(nested[0])), dtype=object)
                for a, xa in enumerate(nested):
                    for c, xc in enumerate(xa):
                        out[a, c] = xc
                return out
        np.save(W_filename, pack(Ws), allow_pickle=True)
        np.save(b_filename, pack(bs), allow_pickle=True)

    def save_action_errors(self, filename, cmpt=0, dtype=np.float64, fmt="%.8e"):
        """(Nbeta, 5) = [beta, A, me, fe, fe/RF] (va_nnet.py:628-650)."""
        self._one()
        savearray = np.zeros((self.Nbeta, 5))
        savearray[:, 0] = self.beta_array
        savearray[:, 1] = self._A[0]
        savearray[:, 2] = self._me[0]
        savearray[:, 3] = self._fe[0]
        savearray[:, 4] = self._fe[0] / (self.RF0 * _alpha_pow(self.alpha, self.beta_array))
        if filename.endswith('.npy'):
            np.save(filename, savearray.astype(dtype))
        else:
            np.savetxt(filename, savearray, fmt=fmt)

    def gen_xtrace(self):
        """kept for API compatibility (va_nnet.py:655-660); nothing is taped here"""
        return np.random.rand(self.NDens + self.NPest)
