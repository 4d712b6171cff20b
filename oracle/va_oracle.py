"""ctypes front-end for oracle/va_oracle.c plus a NumPy twin of the action.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; varanneal_amd/ never does.
Parity status: pinned against tests/golden/*.npz (see va_oracle.h).

The NumPy twin (`numpy_action`) follows the reference's array operations
one-for-one (varanneal/va_ode.py:130-234, 358-380, 404-437; RHS
examples/Lorenz96_D20/Lorenz96_anneal.py:15-16) and is what `scipy_ladder`
drives through scipy.optimize.minimize exactly as _autodiffmin.py:85-86 does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libva_oracle.so")

DISC = {"euler": 0, "trapezoid": 1, "SimpsonHermite": 2, "forwardmap": 3}
RHS = {"lorenz96": 0}


class _Problem(C.Structure):
    _fields_ = [("D", C.c_int32), ("N_model", C.c_int32), ("N_data", C.c_int32),
                ("merr_nskip", C.c_int32), ("L", C.c_int32),
                ("Lidx", C.POINTER(C.c_int32)), ("Y", C.POINTER(C.c_double)),
                ("dt_model", C.c_double),
                ("rm_array", C.POINTER(C.c_double)), ("rm", C.c_double),
                ("rf0_array", C.POINTER(C.c_double)), ("rf0", C.c_double),
                ("NP", C.c_int32), ("NPest", C.c_int32),
                ("Pidx", C.POINTER(C.c_int32)), ("P", C.POINTER(C.c_double)),
                ("disc", C.c_int32), ("rhs", C.c_int32)]


class _Opts(C.Structure):
    _fields_ = [("m", C.c_int32), ("ftol", C.c_double), ("gtol", C.c_double),
                ("maxiter", C.c_int32), ("maxfun", C.c_int64), ("maxls", C.c_int32)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("va_oracle.c", "va_lbfgsb.inc.c", "va_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libva_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        _lib.vao_action_grad.argtypes = [C.POINTER(_Problem), dp, C.c_double, dp, dp, dp, dp]
        _lib.vao_action_grad.restype = C.c_int
        _lib.vao_minimize_lbfgs.argtypes = [C.POINTER(_Problem), dp, C.c_double, C.POINTER(_Opts),
                                            dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int64)]
        _lib.vao_minimize_lbfgs.restype = C.c_int
        _lib.vao_anneal.argtypes = [C.POINTER(_Problem), dp, C.c_double, C.POINTER(C.c_uint16),
                                    C.c_int32, C.POINTER(_Opts), dp, dp, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        _lib.vao_anneal.restype = C.c_int
        _lib.vao_action_grad_batch.argtypes = [C.POINTER(C.POINTER(_Problem)), C.c_int32, dp, C.c_double, dp, dp, dp, dp]
        _lib.vao_action_grad_batch.restype = C.c_int
        _lib.vao_num_threads.restype = C.c_int
    return _lib


_FG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


def lbfgs_generic(fg, x0, opt_args=None, bounds=None, exact=True):
    """The oracle's L-BFGS (restated L-BFGS-B 3.0 unconstrained path + dcsrch, va_oracle.c) on any
    objective `fg(x) -> (f, grad)`: the arbiter for actions other than the ODE one.  `bounds`: list of
    (lo, hi) per variable (None = none): L-BFGS-B itself (generalised Cauchy point + subspace minimisation,
    vao_lbfgsb: SciPy's iterates, what the device runs); exact=False: the active-set form of round 2
    (vao_lbfgs_bounded), kept for comparison only.
    Returns (x, f, status, nit, nfev)."""
    x = np.array(x0, dtype=np.float64)
    n = x.size
    L = lib()
    dp = C.POINTER(C.c_double)
    L.vao_lbfgs_bounded.argtypes = [C.c_int32, dp, _FG, C.c_void_p, dp, dp, C.POINTER(_Opts),
                                    dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.vao_lbfgs_bounded.restype = C.c_int
    lo = hi = None
    if bounds is not None:
        lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
        hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
        assert lo.size == n

    def cb(ctx, xp, fp, gp):
        xv = np.ctypeslib.as_array(xp, shape=(n,))
        f, g = fg(xv.copy())
        fp[0] = float(f)
        np.ctypeslib.as_array(gp, shape=(n,))[:] = g
        return 0
    o = Problem._opts(opt_args)
    A, st, nit, nfev = C.c_double(), C.c_int32(), C.c_int32(), C.c_int64()
    if exact:
        if lo is None:
            lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
        L.vao_lbfgsb.argtypes = L.vao_lbfgs_bounded.argtypes
        L.vao_lbfgsb.restype = C.c_int
        rc = L.vao_lbfgsb(n, _dp(x), _FG(cb), None, _dp(lo), _dp(hi), C.byref(o), C.byref(A), C.byref(st), C.byref(nit), C.byref(nfev))
        if rc:
            raise ValueError("vao_lbfgsb rc=%d" % rc)
        return x, A.value, st.value, nit.value, nfev.value
    rc = L.vao_lbfgs_bounded(n, _dp(x), _FG(cb), None, _dp(lo) if lo is not None else None,
                             _dp(hi) if hi is not None else None, C.byref(o), C.byref(A), C.byref(st), C.byref(nit),
                             C.byref(nfev))
    if rc:
        raise ValueError("vao_lbfgs_bounded rc=%d" % rc)
    return x, A.value, st.value, nit.value, nfev.value


def set_num_threads(n):
    lib().vao_set_num_threads(int(n))


def num_threads():
    """threads the batch evaluation runs on (OpenMP; OMP_NUM_THREADS or every host core)"""
    return lib().vao_num_threads()


def action_grad_batch(problems, XP, rf_scale=1.0, want_grad=True):
    """(A, me, fe, grad) of len(problems) seeds, one per OpenMP thread (vao_action_grad_batch)."""
    n = len(problems)
    XP = np.ascontiguousarray(XP, dtype=np.float64)
    assert XP.shape == (n, problems[0].n_var)
    arr = (C.POINTER(_Problem) * n)(*[C.pointer(p._s) for p in problems])
    A = np.empty(n); me = np.empty(n); fe = np.empty(n)
    g = np.empty_like(XP) if want_grad else None
    rc = lib().vao_action_grad_batch(arr, n, _dp(XP), float(rf_scale), _dp(A), _dp(me), _dp(fe),
                                     _dp(g) if want_grad else None)
    if rc:
        raise ValueError("vao_action_grad_batch rc=%d" % rc)
    return A, me, fe, g


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Problem(object):
    """Frozen (Y, Lidx, RM, RF0, dt, disc, P) -- one seed."""

    def __init__(self, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx,
                 disc="trapezoid", rhs="lorenz96", merr_nskip=1):
        self.D, self.N = int(D), int(N_model)
        self.Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.N_data, self.L = self.Y.shape
        self.Lidx = np.ascontiguousarray(Lidx, dtype=np.int32)
        self.P = np.ascontiguousarray(P, dtype=np.float64).copy()
        self.Pidx = np.ascontiguousarray(Pidx, dtype=np.int32)
        self.NP, self.NPest = len(self.P), len(self.Pidx)
        self.dt, self.disc, self.rhs, self.nskip = float(dt_model), disc, rhs, int(merr_nskip)
        self.RM, self.RF0 = RM, RF0
        self._rm_arr = self._rf_arr = None
        s = _Problem()
        s.D, s.N_model, s.N_data, s.merr_nskip, s.L = self.D, self.N, self.N_data, self.nskip, self.L
        s.Lidx = self.Lidx.ctypes.data_as(C.POINTER(C.c_int32))
        s.Y = _dp(self.Y)
        s.dt_model = self.dt
        if isinstance(RM, np.ndarray):
            self._rm_arr = np.ascontiguousarray(RM, dtype=np.float64)
            assert self._rm_arr.shape == (self.N_data, self.L)
            s.rm_array, s.rm = _dp(self._rm_arr), 0.0
        else:
            s.rm_array, s.rm = None, float(RM)
        if isinstance(RF0, np.ndarray):
            self._rf_arr = np.ascontiguousarray(RF0, dtype=np.float64)
            assert self._rf_arr.shape == (self.N - 1, self.D)
            s.rf0_array, s.rf0 = _dp(self._rf_arr), 0.0
        else:
            s.rf0_array, s.rf0 = None, float(RF0)
        s.NP, s.NPest = self.NP, self.NPest
        s.Pidx = self.Pidx.ctypes.data_as(C.POINTER(C.c_int32))
        s.P = _dp(self.P)
        s.disc, s.rhs = DISC[disc], RHS[rhs]
        self._s = s

    @property
    def n_var(self):
        return self.N * self.D + self.NPest

    def action_grad(self, XP, rf_scale=1.0, want_grad=True):
        XP = np.ascontiguousarray(XP, dtype=np.float64)
        assert XP.shape == (self.n_var,)
        A, me, fe = C.c_double(), C.c_double(), C.c_double()
        g = np.empty(self.n_var) if want_grad else None
        rc = lib().vao_action_grad(C.byref(self._s), _dp(XP), float(rf_scale), C.byref(A),
                                   C.byref(me), C.byref(fe), _dp(g) if want_grad else None)
        if rc:
            raise ValueError("vao_action_grad rc=%d" % rc)
        return A.value, me.value, fe.value, g

    @staticmethod
    def _opts(opt_args):
        o = dict(opt_args or {})
        return _Opts(int(o.get("maxcor", 10)), float(o.get("ftol", 2.2204460492503131e-09)),
                     float(o.get("gtol", 1e-5)), int(min(o.get("maxiter", 15000), 2**31 - 1)),
                     int(o.get("maxfun", 15000)), int(o.get("maxls", 20)))

    def minimize_lbfgs(self, XP0, rf_scale, opt_args=None, bounds=None, exact=True):
        if bounds is not None:       # exact: L-BFGS-B itself (vao_lbfgsb); else the active-set form (vao_lbfgs_bounded)
            fg = lambda z: (lambda r: (r[0], r[3]))(self.action_grad(z, rf_scale))
            return lbfgs_generic(fg, XP0, opt_args, bounds, exact=exact)
        x = np.array(XP0, dtype=np.float64)
        o = self._opts(opt_args)
        A, st, nit, nfev = C.c_double(), C.c_int32(), C.c_int32(), C.c_int64()
        rc = lib().vao_minimize_lbfgs(C.byref(self._s), _dp(x), float(rf_scale), C.byref(o),
                                      C.byref(A), C.byref(st), C.byref(nit), C.byref(nfev))
        if rc:
            raise ValueError("vao_minimize_lbfgs rc=%d" % rc)
        return x, A.value, st.value, nit.value, nfev.value

    def anneal(self, XP0, alpha, beta_array, opt_args=None):
        beta = np.ascontiguousarray(beta_array, dtype=np.uint16)
        nb = len(beta)
        o = self._opts(opt_args)
        XP0 = np.ascontiguousarray(XP0, dtype=np.float64)
        minpaths = np.zeros((nb, self.N * self.D + self.NP))
        ame = np.zeros((nb, 3))
        st = np.zeros(nb, dtype=np.int32); nit = np.zeros(nb, dtype=np.int32)
        nfev = np.zeros(nb, dtype=np.int64)
        rc = lib().vao_anneal(C.byref(self._s), _dp(XP0), float(alpha),
                              beta.ctypes.data_as(C.POINTER(C.c_uint16)), nb, C.byref(o),
                              _dp(minpaths), _dp(ame), st.ctypes.data_as(C.POINTER(C.c_int32)),
                              nit.ctypes.data_as(C.POINTER(C.c_int32)),
                              nfev.ctypes.data_as(C.POINTER(C.c_int64)))
        if rc:
            raise ValueError("vao_anneal rc=%d" % rc)
        return dict(minpaths=minpaths, A=ame[:, 0], me=ame[:, 1], fe=ame[:, 2], status=st,
                    nit=nit, nfev=nfev)

    # ------------------------------------------------------------------
    # NumPy twin: the reference's array ops, one for one.
    def numpy_action(self, XP, rf_scale=1.0):
        N, D = self.N, self.D
        x = np.reshape(XP[:N * D], (N, D))
        p = np.array(self.P, dtype=XP.dtype)
        p[self.Pidx] = XP[N * D:]
        diff = x[::self.nskip, self.Lidx] - self.Y                       # va_ode.py:143
        if self._rm_arr is not None:
            me = np.sum(self._rm_arr * diff * diff)
        else:
            me = self.RM * np.sum(diff * diff)
        me = me / (self.L * self.N_data)
        f = lambda xx: (np.roll(xx, 1, 1) * (np.roll(xx, -1, 1) - np.roll(xx, 2, 1)) - xx + p[0])
        RF = (self._rf_arr if self._rf_arr is not None else self.RF0) * rf_scale
        arr = self._rf_arr is not None
        dt = self.dt
        if self.disc == "SimpsonHermite":
            fn, fmid, fnp1 = f(x[:-2:2]), f(x[1:-1:2]), f(x[2::2])
            v1 = (fn + 4.0 * fmid + fnp1) * (2.0 * dt) / 6.0
            v2 = (x[:-2:2] + x[2::2]) / 2.0 + (fn - fnp1) * (2.0 * dt) / 8.0
            d1 = x[2::2] - x[:-2:2] - v1
            d2 = x[1::2] - v2
            if arr:
                fe = np.sum(RF[::2] * d1 * d1) + np.sum(RF[1::2] * d2 * d2)
            else:
                fe = RF * np.sum(d1 * d1 + d2 * d2)
        else:
            if self.disc == "trapezoid":
                d = x[1:] - x[:-1] - dt * (f(x[:-1]) + f(x[1:])) / 2.0
            elif self.disc == "euler":
                d = x[1:] - x[:-1] - dt * f(x[:-1])
            else:
                d = x[1:] - f(x[:-1])
            fe = np.sum(RF * d * d) if arr else RF * np.sum(d * d)
        fe = fe / (D * (N - 1))
        return me + fe, me, fe


def numpy_action_generic(f, XP, D, N, Y, Lidx, dt, RM, RF, NP, Pidx, P, disc, t_model=None, stim=None,
                         nskip=1):
    """The reference's action (va_ode.py:130-234, 341-454) for an arbitrary user `f`, array op
    for array op, including the stimulus tuple convention (:345-375).  Type-polymorphic, so a
    complex XP gives complex-step derivatives.  RM: scalar, (N_data,L) or (N_data,L,L); RF: scalar, (N-1,D) or (N-1,D,D).
    P of shape (N, NP) = time-dependent parameters (trapezoid / SimpsonHermite as upstream)."""
    x = np.reshape(XP[:N * D], (N, D))
    p = np.array(P, dtype=XP.dtype)
    tdp = p.ndim == 2                       # time-dependent parameters: P is (N, NP) (va_ode.py:170-188)
    if tdp:
        p[:, list(Pidx)] = np.reshape(XP[N * D:], (N, len(Pidx)))
    else:
        p[list(Pidx)] = XP[N * D:]
    t = np.zeros(N) if t_model is None else np.asarray(t_model)
    diff = x[::nskip, list(Lidx)] - Y
    if isinstance(RM, np.ndarray) and RM.ndim == 3:              # full matrices (va_ode.py:149-152)
        me = sum(np.dot(diff[i], np.dot(RM[i], diff[i])) for i in range(Y.shape[0])) / (len(Lidx) * Y.shape[0])
    else:
        me = (np.sum(RM * diff * diff) if isinstance(RM, np.ndarray) else RM * np.sum(diff * diff)) / (len(Lidx) * Y.shape[0])
    pp = (lambda sl: p[sl]) if tdp else (lambda sl: p)            # f sees the rows' own parameters
    arg = pp if stim is None else (lambda sl: (pp(sl), stim[sl]))
    arr = isinstance(RF, np.ndarray)
    full = arr and RF.ndim == 3              # (N-1, D, D) matrices: diff_n . (RF_n . diff_n), va_ode.py:211-217
    quad = lambda R, d: sum(np.dot(d[i], np.dot(R[i], d[i])) for i in range(d.shape[0]))
    if disc == "SimpsonHermite":
        a, m_, b = slice(None, -2, 2), slice(1, -1, 2), slice(2, None, 2)
        fn, fmid, fnp1 = f(t[a], x[a], arg(a)), f(t[m_], x[m_], arg(m_)), f(t[b], x[b], arg(b))
        v1 = (fn + 4.0 * fmid + fnp1) * (2.0 * dt) / 6.0
        v2 = (x[a] + x[b]) / 2.0 + (fn - fnp1) * (2.0 * dt) / 8.0
        d1 = x[2::2] - x[:-2:2] - v1
        d2 = x[1::2] - v2
        if full:
            fe = quad(RF[::2], d1) + quad(RF[1::2], d2)
        else:
            fe = (np.sum(RF[::2] * d1 * d1) + np.sum(RF[1::2] * d2 * d2)) if arr else RF * np.sum(d1 * d1 + d2 * d2)
    else:
        a, b = slice(None, -1), slice(1, None)
        if disc == "trapezoid":
            d = x[1:] - x[:-1] - dt * (f(t[a], x[a], arg(a)) + f(t[b], x[b], arg(b))) / 2.0
        elif disc == "euler":
            d = x[1:] - x[:-1] - dt * f(t[a], x[a], arg(a))
        else:
            d = x[1:] - f(t[a], x[a], arg(a))
        # (full matrices: upstream's branch for these discretisations, va_ode.py:218-222, contracts RF[i] with
        # the whole diff array -- a slip; the contraction meant, and the one Simpson-Hermite uses, is per row)
        fe = quad(RF, d) if full else (np.sum(RF * d * d) if arr else RF * np.sum(d * d))
    fe = fe / (D * (N - 1))
    return me + fe, me, fe


def complex_step_grad(fun, XP, h=1e-30):
    z = np.asarray(XP, dtype=np.complex128).copy()
    g = np.empty(len(z))
    for i in range(len(z)):
        z[i] = complex(XP[i], h)
        g[i] = fun(z)[0].imag / h
        z[i] = XP[i]
    return g


def scipy_ladder(pb, XP0, alpha, beta_array, opt_args):
    """Reference control flow (va_ode.py:707-789 + _autodiffmin.py:85-86) with
    the C oracle as A_gradA_taped and SciPy's own L-BFGS-B as the minimiser."""
    import scipy.optimize as opt
    beta = np.array(beta_array, dtype=np.uint16)
    ND = pb.N * pb.D
    xp = np.array(XP0, dtype=np.float64)
    out = dict(A=[], me=[], fe=[], P=[], nit=[], nfev=[], status=[], minpaths=[])
    for b in beta:
        rf = alpha ** b

        def fg(z, rf=rf):
            A, _, _, g = pb.action_grad(z, rf)
            return A, g
        res = opt.minimize(fg, xp, method="L-BFGS-B", jac=True, options=opt_args)
        xp = res.x.copy()
        pb.P[pb.Pidx] = xp[ND:]
        _, me, fe, _ = pb.action_grad(xp, rf, want_grad=False)
        out["A"].append(res.fun); out["me"].append(me); out["fe"].append(fe)
        out["P"].append(pb.P.copy()); out["nit"].append(res.nit); out["nfev"].append(res.nfev)
        out["status"].append(res.status)
        out["minpaths"].append(np.append(xp[:ND], pb.P))
    return {k: np.array(v) for k, v in out.items()}
