import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import numpy as np
from _util import load_npz_cases
from varanneal_amd import _capi, twin
c = load_npz_cases("nnet.npz")["g7_small_tanh_ladder"]
s, M = c["structure"], int(c["M"])
X0, P0, Pidx = twin.nnet_initial_guess(s, M, 1)
X = X0.reshape(M, -1); X[:, :s[0]] = c["din"]; X[:, -s[-1]:] = c["dout"]
XP0 = np.append(X0, P0[Pidx])[None, :]
Lidx = [np.arange(s[0]), np.arange(s[-1])]
def mk(small):
    os.environ["VA_NNET_SMALL"] = "1" if small else "0"
    return _capi.NnetProblem(1, s, c["din"], c["dout"], Lidx, float(c["RM"]), float(c["RF0"]), P0[None, :], Pidx, act="tanh")
a, b = mk(True), mk(False)
for rf in (1.0, 1.5 ** 2):
    Aa, _, _, ga = a.action_grad(XP0, rf); Ab, _, _, gb = b.action_grad(XP0, rf)
    print("eval", rf, Aa, Ab, np.abs(ga - gb).max() / np.abs(gb).max())
    for k in (1, 2, 3, 5, 10, 20, 40, 1000):
        o = {'gtol': 1e-12, 'ftol': 1e-12, 'maxfun': 1000000, 'maxiter': k}
        ra = a.minimize_lbfgs(XP0, rf, o); rb = b.minimize_lbfgs(XP0, rf, o)
        print(k, "small A=%.15e nit=%d nfev=%d st=%d | tiled A=%.15e nit=%d nfev=%d st=%d | dx=%.2e" % (
            ra["A"][0], ra["nit"][0], ra["nfev"][0], ra["status"][0], rb["A"][0], rb["nit"][0], rb["nfev"][0], rb["status"][0],
            np.abs(ra["x"] - rb["x"]).max()))
