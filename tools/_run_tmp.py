import numpy as np, sys
sys.path.insert(0, '.')
from varanneal_amd import _capi, twin
D, N, B = 200, 5001, 64
Lidx = list(range(0, D, 5))
t, Y, _, _ = twin.make_twin(D, N, Lidx=Lidx)
rng = np.random.RandomState(0)
XP = np.concatenate([8.0 * rng.rand(B, N * D) - 4.0, 8.17 + 0.1 * rng.randn(B, 1)], axis=1)
P = XP[:, N * D:].copy()
for disc in ("SimpsonHermite", "trapezoid"):
    for ek in (5, 3):
        pr = _capi.Problem(B, D, N, Y, Lidx, 0.025, 4.0, 4e-6, P, [0], disc=disc, eval_kernel=ek)
        out = pr.action_grad(XP, 1.5 ** 20)
        pr.eval_timed(1.5 ** 20, 20)
        us = min(pr.eval_timed(1.5 ** 20, 50) for _ in range(3)) / 50 * 1e3
        print(disc, ek, pr.info()["eval_kernel"], "us=%.1f" % us, "frac=%.3f" % (16.0 * B * N * D / (us * 1e-6) / 8e12), "A0=%.15g" % out[0][0], flush=True)
        pr.close()
