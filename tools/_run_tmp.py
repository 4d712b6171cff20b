import numpy as np, sys
sys.path.insert(0, '.')
from varanneal_amd import _capi, twin
D, N, B = 200, 5000, 64
Lidx = list(range(0, D, 5))
t, Y, _, _ = twin.make_twin(D, N, Lidx=Lidx)
rng = np.random.RandomState(0)
XP = np.concatenate([8.0 * rng.rand(B, N * D) - 4.0, 8.17 + 0.1 * rng.randn(B, 1)], axis=1)
P = XP[:, N * D:].copy()
RF = 4e-6 * (0.5 + rng.rand(N - 1, D))
RMa = 4.0 * (0.5 + rng.rand(N, len(Lidx)))
for name, RM, RFw, nskip in (("scalar", 4.0, 4e-6, 1), ("RF (N-1,D)", 4.0, RF, 1), ("RM (N,L) + RF (N-1,D)", RMa, RF, 1), ("nskip=2", 4.0, 4e-6, 2)):
    Nn = N if nskip == 1 else ((N - 1) // nskip) * nskip + 1
    Yn = Y[:Nn:nskip]
    XPn = XP if Nn == N else np.concatenate([XP[:, :Nn * D], P], axis=1)
    for ek in (5, 3):
        pr = _capi.Problem(B, D, Nn, Yn, Lidx, 0.025, RM, RFw, P, [0], disc="trapezoid", eval_kernel=ek, merr_nskip=nskip)
        out = pr.action_grad(XPn, 1.5 ** 20)
        pr.eval_timed(1.5 ** 20, 20)
        us = min(pr.eval_timed(1.5 ** 20, 50) for _ in range(3)) / 50 * 1e3
        nb = 16.0
        print("%-24s kernel %d: %.1f us, %.3f of 8 TB/s at %d B per element, A0=%.15g" % (name, pr.info()["eval_kernel"], us, nb * B * Nn * D / (us * 1e-6) / 8e12, nb, out[0][0]), flush=True)
        pr.close()
