"""The reference's `ADmin` mixin surface (varanneal/_autodiffmin.py:16-160) on the device.

`va_ode.Annealer` and `va_nnet.Annealer` inherit their minimisers from ADmin upstream
(va_ode.py:41,43; va_nnet.py:41,43); user code occasionally calls those methods directly
(`A_gradA_taped`, `min_lbfgs_scipy`, ...).  Same names and return values here: there is no tape,
`A` and its gradient come from one kernel launch at the current RF, `min_lbfgs_scipy` is the
device-resident L-BFGS (bounds=None) and the CG / TNC / bounded routes run SciPy on the host
around the device evaluator exactly as upstream calls it."""
import numpy as np


def alpha_pow(alpha, beta):
    """alpha**beta as the reference's legacy NumPy evaluated it (va_ode.py:650,782): in float64.
    beta_array is cast to uint16 upstream (va_ode.py:644); under NumPy 2 an integer alpha raised to a
    uint16 array stays uint16 and wraps (2**16 -> 0), which would switch the model term off."""
    return np.power(np.float64(alpha), np.asarray(beta, dtype=np.float64))


class HIPmin(object):
    def tape_A(self, xtrace=None):
        """_autodiffmin.py:32-49: nothing to record (RF is a kernel argument)."""
        self.taped = True

    def _one_seed(self):
        if getattr(self, "_pb", None) is None:
            raise RuntimeError("anneal_init() has not been called")
        if self.B != 1:
            raise ValueError("the ADmin-style single-vector methods need a single-seed Annealer")

    def _rf_now(self):
        return float(self._rf_scale[self.betaidx])

    def A_taped(self, XP):
        """_autodiffmin.py:51-52"""
        return self._eval(XP, False)[0]

    def gradA_taped(self, XP):
        """_autodiffmin.py:54-55"""
        return self._eval(XP, True)[3]

    def A_gradA_taped(self, XP):
        """_autodiffmin.py:57-58: (A, grad A) at the current RF."""
        A, me, fe, g = self._eval(XP, True)
        return A, g

    def jacA_taped(self, XP):
        raise NotImplementedError("vector action / Jacobian (Levenberg-Marquardt) is dead code upstream "
                                  "(_autodiffmin.py:60-64, :145-160)")

    A_jacaA_taped = jacA_taped

    def hessianA_taped(self, XP):
        raise NotImplementedError("second derivatives (_autodiffmin.py:66-67) are not part of the annealing path")

    def min_lbfgs_scipy(self, XP0, xtrace=None):
        """_autodiffmin.py:72-95 -> (XPmin, Amin, status); SciPy status codes 0 / 1 / 2."""
        self._one_seed()
        XP0 = np.asarray(XP0, dtype=np.float64)[None, :]
        if self.bounds is None:
            r = self._pb.minimize_lbfgs(XP0, self._rf_now(), self.opt_args)
            return r["x"][0], float(r["A"][0]), int(r["status"][0])
        return self._host_minimise('L-BFGS-B', XP0[0])

    def min_cg_scipy(self, XP0, xtrace=None):
        """_autodiffmin.py:97-119"""
        self._one_seed()
        return self._host_minimise('CG', np.asarray(XP0, dtype=np.float64))

    def min_tnc_scipy(self, XP0, xtrace=None):
        """_autodiffmin.py:121-143"""
        self._one_seed()
        return self._host_minimise('TNC', np.asarray(XP0, dtype=np.float64))

    def min_lm_scipy(self, XP0, xtrace=None):
        raise NotImplementedError("min_lm_scipy is unfinished upstream (_autodiffmin.py:145-160)")

    def _host_minimise(self, meth, XP0):
        import scipy.optimize as opt
        rf = self._rf_now()

        def fg(z):
            A, me, fe, g = self._pb.action_grad(z[None, :], rf)
            return A[0], g[0]
        kw = dict(method=meth, jac=True, options=self.opt_args)
        if meth != 'CG':
            kw["bounds"] = self.bounds
        res = opt.minimize(fg, XP0, **kw)
        return res.x, float(res.fun), int(res.status)
