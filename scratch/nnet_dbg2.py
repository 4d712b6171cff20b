import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import numpy as np, scipy.optimize as opt
from _util import load_npz_cases
from varanneal_amd import _capi, twin, va_nnet
c = load_npz_cases("nnet.npz")["g7_twin_ladder"]
OPTS = {'gtol': 1.0e-12, 'ftol': 1.0e-12, 'maxfun': 1000000, 'maxiter': 1000000}
a = va_nnet.Annealer()
a.set_structure(c["structure"]); a.set_activation(twin.sigmoid); a.set_input_data(c["din"]); a.set_output_data(c["dout"])
a.anneal(c["X0"].copy(), c["P0"].copy(), float(c["alpha"]), c["beta"], float(c["RM"]), float(c["RF0"]),
         list(c["Pidx"]), Lidx=[np.arange(10), np.arange(10)], opt_args=OPTS, verbose=False)
print("nit", a.nit_array); print("nfev", a.nfev_array); print("flags", a.exitflags)
pr = a._pb
for k in (26, 27, 28, 29):
    xp0 = a._xp0(k)
    rf = float(a._rf_scale[k])
    def fg(z):
        A, me, fe, g = pr.action_grad(z[None, :], rf)
        return A[0], g[0]
    A0, g0 = fg(xp0[0])
    rs = opt.minimize(fg, xp0[0], method='L-BFGS-B', jac=True, options=OPTS)
    r = pr.minimize_lbfgs(xp0, rf, OPTS)
    print(k, "rf=%.3e start A=%.10e |g|max=%.3e | dev A=%.12e nit=%d nfev=%d st=%d | scipy A=%.12e nit=%d nfev=%d st=%d %s" % (
        rf, A0, np.abs(g0).max(), r["A"][0], r["nit"][0], r["nfev"][0], r["status"][0], rs.fun, rs.nit, rs.nfev, rs.status, rs.message))
    for mi in (1, 2, 3):
        o = dict(OPTS); o['maxiter'] = mi
        r = pr.minimize_lbfgs(xp0, rf, o); rs = opt.minimize(fg, xp0[0], method='L-BFGS-B', jac=True, options=o)
        print("    maxiter", mi, "dev A=%.15e nfev=%d st=%d | scipy A=%.15e nfev=%d" % (r["A"][0], r["nfev"][0], r["status"][0], rs.fun, rs.nfev))
