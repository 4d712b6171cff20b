import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import numpy as np, scipy.optimize as opt
from _util import load_npz_cases
from varanneal_amd import _capi, twin
c = load_npz_cases("nnet.npz")["g7_twin_ladder"]
s = c["structure"]; M = 2
X0 = c["X0"].copy().reshape(2, 200); X0[:, :10] = c["din"]; X0[:, 190:] = c["dout"]
XP0 = np.append(X0.ravel(), c["P0"][c["Pidx"]])
pr = _capi.NnetProblem(1, s, c["din"], c["dout"], [np.arange(10), np.arange(10)], float(c["RM"]), float(c["RF0"]),
                       c["P0"][None, :], c["Pidx"])
def fg(z):
    A, me, fe, g = pr.action_grad(z[None, :], 1.0)
    return A[0], g[0]
for k in (1, 2, 3, 4, 5, 6, 8, 12, 16, 20, 30):
    o = {'gtol': 1e-12, 'ftol': 1e-12, 'maxfun': 1000000, 'maxiter': k}
    r = pr.minimize_lbfgs(XP0[None, :], 1.0, o)
    rs = opt.minimize(fg, XP0, method='L-BFGS-B', jac=True, options=o)
    print(k, "dev A=%.17e nit=%d nfev=%d st=%d | scipy A=%.17e nit=%d nfev=%d st=%d | dx=%.3e" % (
        r["A"][0], r["nit"][0], r["nfev"][0], r["status"][0], rs.fun, rs.nit, rs.nfev, rs.status,
        np.abs(r["x"][0] - rs.x).max()))
