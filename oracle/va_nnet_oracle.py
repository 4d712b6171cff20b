"""CPU restatement of the reference's feed-forward-network action (TEST INFRASTRUCTURE).

Follows varanneal/va_nnet.py one array operation at a time:
  me_gaussian   va_nnet.py:117-173   (scalar RM or RM = [RM_in, RM_out])
  fe_gaussian   va_nnet.py:175-255   (flat parameter slicing :194-207; per example m and
                                      layer n: x_{n+1} - f(x_n, W_n, b_n), :225-246;
                                      normalisation (NDnet - structure[0]) * M, :255)
  anneal_init   va_nnet.py:288-450   (Lidx pair, init_to_data :423-430, XP packing :440)
  anneal_step   va_nnet.py:452-523   (warm start, parameter write-back, RF update)
plus a hand-derived adjoint (`action_grad`) used where the reference would call ADOL-C.
Pinned by tests/golden/nnet.npz (generated from the reference itself by
oracle/gen_golden_nnet.py; gradients there are complex-step derivatives through the
reference's own A).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
import this file.
"""
import numpy as np

ACTS = {
    # name -> (g(z), g'(z) expressed through a = g(z) and z)
    "sigmoid": (lambda z: 1.0 / (1.0 + np.exp(-z)), lambda a, z: a * (1.0 - a)),
    "tanh": (lambda z: np.tanh(z), lambda a, z: 1.0 - a * a),
    "linear": (lambda z: z, lambda a, z: np.ones_like(z)),
    "softplus": (lambda z: np.log(1.0 + np.exp(z)), lambda a, z: 1.0 / (1.0 + np.exp(-z))),
    "relu": (lambda z: z * (z.real > 0), lambda a, z: 1.0 * (z > 0)),
}


class NnetProblem(object):
    def __init__(self, structure, data_in, data_out, Lidx, RM, RF0, P, Pidx, act="sigmoid"):
        self.s = [int(v) for v in structure]
        self.N = len(self.s)
        self.M = int(np.shape(data_in)[0])
        self.NDnet = int(sum(self.s)); self.NDens = self.NDnet * self.M
        self.din = np.asarray(data_in, dtype=np.float64); self.dout = np.asarray(data_out, dtype=np.float64)
        self.Lin = np.asarray(Lidx[0], dtype=int); self.Lout = np.asarray(Lidx[1], dtype=int)
        self.Ltot = len(self.Lin) + len(self.Lout)
        self.RM = RM; self.RF0 = float(RF0)
        self.P = np.array(P, dtype=np.float64); self.NP = self.P.size
        self.Pidx = np.asarray(Pidx, dtype=int); self.NPest = self.Pidx.size
        self.act = act
        self.off = np.concatenate([[0], np.cumsum(self.s)]).astype(int)
        self.woff, self.boff, o = [], [], 0
        for n in range(self.N - 1):
            self.woff.append(o); o += self.s[n + 1] * self.s[n]
            self.boff.append(o); o += self.s[n + 1]
        assert o == self.NP, "P has %d entries, structure needs %d" % (self.NP, o)

    @property
    def n_var(self):
        return self.NDens + self.NPest

    def _rm(self):
        """(RM_in, RM_out): scalars, or the two matrices of va_nnet.py:136-139 (diff . (RM . diff) per example)"""
        RM = self.RM
        if isinstance(RM, (list, tuple)) or (isinstance(RM, np.ndarray) and RM.ndim > 0):
            if np.ndim(RM[0]) == 2:
                return np.asarray(RM[0], dtype=np.float64), np.asarray(RM[1], dtype=np.float64)
            if np.shape(RM) != (2,):
                raise NotImplementedError("RM: scalar, [RM_in, RM_out] or two matrices")
            return float(RM[0]), float(RM[1])
        return float(RM), float(RM)

    def _acts(self):
        """(g(z), g'(z) as a function of (a, z)): a built-in name or such a pair"""
        return ACTS[self.act] if isinstance(self.act, str) else self.act

    @staticmethod
    def _quad(R, d):
        """sum over examples of d_m . (R . d_m) for a matrix R, R * sum d^2 for a scalar"""
        return np.sum(d * (d @ np.asarray(R).T)) if np.ndim(R) == 2 else R * np.sum(d * d)

    def _unpack(self, XP):
        XP = np.asarray(XP)
        X = XP[:self.NDens].reshape(self.M, self.NDnet)
        p = np.array(self.P, dtype=XP.dtype)
        if self.NPest:
            p[self.Pidx] = XP[self.NDens:]
        return X, p

    def action(self, XP, rf_scale=1.0):
        """(A, me, fe): vectorised over the M examples, otherwise the reference's arithmetic;
        accepts complex XP (complex-step differentiation)."""
        X, p = self._unpack(XP)
        g, _ = self._acts()
        rmi, rmo = self._rm()
        xin = X[:, :self.s[0]][:, self.Lin]
        xout = X[:, self.NDnet - self.s[-1]:][:, self.Lout]
        me = (self._quad(rmi, xin - self.din) + self._quad(rmo, xout - self.dout)) / float(self.Ltot * self.M)
        fe = 0.0
        for n in range(self.N - 1):
            W = p[self.woff[n]:self.boff[n]].reshape(self.s[n + 1], self.s[n])
            b = p[self.boff[n]:self.boff[n] + self.s[n + 1]]
            xn = X[:, self.off[n]:self.off[n + 1]]
            xn1 = X[:, self.off[n + 1]:self.off[n + 2]]
            r = xn1 - g(xn @ W.T + b)
            fe = fe + np.sum(r * r)
        fe = self.RF0 * rf_scale * fe / float((self.NDnet - self.s[0]) * self.M)
        return me + fe, me, fe

    def action_grad(self, XP, rf_scale=1.0):
        """(A, me, fe, grad) with the adjoint written out by hand (real XP only)."""
        X, p = self._unpack(np.asarray(XP, dtype=np.float64))
        g, dg = self._acts()
        rmi, rmo = self._rm()
        cme = 1.0 / float(self.Ltot * self.M)
        cfe = self.RF0 * rf_scale / float((self.NDnet - self.s[0]) * self.M)
        gX = np.zeros_like(X); gp = np.zeros(self.NP)
        din = X[:, :self.s[0]][:, self.Lin] - self.din
        dout = X[:, self.NDnet - self.s[-1]:][:, self.Lout] - self.dout
        me = cme * (self._quad(rmi, din) + self._quad(rmo, dout))
        dme = lambda R, d: d @ (np.asarray(R) + np.asarray(R).T) if np.ndim(R) == 2 else 2.0 * R * d
        np.add.at(gX, (slice(None), self.Lin), cme * dme(rmi, din))
        np.add.at(gX, (slice(None), self.NDnet - self.s[-1] + self.Lout), cme * dme(rmo, dout))
        fe = 0.0
        for n in range(self.N - 1):
            W = p[self.woff[n]:self.boff[n]].reshape(self.s[n + 1], self.s[n])
            b = p[self.boff[n]:self.boff[n] + self.s[n + 1]]
            xn = X[:, self.off[n]:self.off[n + 1]]
            xn1 = X[:, self.off[n + 1]:self.off[n + 2]]
            z = xn @ W.T + b
            a = g(z)
            r = xn1 - a
            fe += np.sum(r * r)
            q = 2.0 * cfe * r
            delta = -q * dg(a, z)                       # dA/dz
            gX[:, self.off[n + 1]:self.off[n + 2]] += q
            gX[:, self.off[n]:self.off[n + 1]] += delta @ W
            gp[self.woff[n]:self.boff[n]] = (delta.T @ xn).ravel()
            gp[self.boff[n]:self.boff[n] + self.s[n + 1]] = delta.sum(axis=0)
        fe *= cfe
        grad = np.concatenate([gX.ravel(), gp[self.Pidx]])
        return me + fe, me, fe, grad

    def minimize_lbfgs(self, XP0, rf_scale, opt_args=None):
        """The arbiter of trajectory-level parity: this action under the
        oracle's OWN L-BFGS (va_oracle.c: vao_lbfgs_generic -- the optimiser the device restates), from
        a given start point.  Returns (x, A, status, nit, nfev)."""
        import va_oracle

        def fg(x):
            A, me, fe, g = self.action_grad(x, rf_scale)
            return A, g
        return va_oracle.lbfgs_generic(fg, XP0, opt_args)

    def reference_loop_action(self, XP, rf_scale=1.0, f=None):
        """The reference's own double loop (va_nnet.py:209-255) with a user activation
        f(x, W, b): the slow form, used to check the vectorised one."""
        X, p = self._unpack(XP)
        fe = 0.0
        for m in range(self.M):
            for n in range(self.N - 1):
                W = p[self.woff[n]:self.boff[n]].reshape(self.s[n + 1], self.s[n])
                b = p[self.boff[n]:self.boff[n] + self.s[n + 1]]
                d = X[m, self.off[n + 1]:self.off[n + 2]] - f(X[m, self.off[n]:self.off[n + 1]], W, b)
                fe = fe + self.RF0 * rf_scale * np.sum(d * d)
        return fe / float((self.NDnet - self.s[0]) * self.M)

    def scipy_ladder(self, X0, alpha, beta_array, opt_args, init_to_data=True):
        """anneal()/anneal_step() of the reference (va_nnet.py:267-523) around SciPy."""
        import scipy.optimize as opt
        X0 = np.array(X0, dtype=np.float64)
        if init_to_data:
            Xv = X0.reshape(self.M, self.NDnet)
            Xv[:, :self.s[0]][:, self.Lin] = self.din
            Xv[:, self.NDnet - self.s[-1] + self.Lout] = self.dout
        xp = np.append(X0, self.P[self.Pidx])
        out = dict(A=[], me=[], fe=[], nit=[], nfev=[], status=[], minpaths=[])
        for beta in beta_array:
            rf = alpha ** float(beta)
            res = opt.minimize(lambda z: self.action_grad(z, rf)[::3], xp, method='L-BFGS-B', jac=True,
                               options=opt_args)
            xp = res.x
            self.P[self.Pidx] = xp[self.NDens:]
            A, me, fe = self.action(xp, rf)
            out["A"].append(res.fun); out["me"].append(me); out["fe"].append(fe)
            out["nit"].append(res.nit); out["nfev"].append(res.nfev); out["status"].append(res.status)
            out["minpaths"].append(np.append(xp[:self.NDens], self.P))
        return {k: np.array(v) for k, v in out.items()}


def complex_step_grad(fun, XP, h=1e-30):
    XP = np.asarray(XP, dtype=np.float64)
    z = XP.astype(np.complex128)
    g = np.empty(XP.size)
    for i in range(XP.size):
        z[i] = complex(XP[i], h)
        g[i] = np.imag(fun(z)) / h
        z[i] = XP[i]
    return g
