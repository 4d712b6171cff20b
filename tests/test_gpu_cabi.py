"""GPU: the C-ABI driven by a host program in plain C (tests/c_client/va_client.c, compiled here with gcc against
include/varanneal_amd.h and linked with libvaranneal_amd.so) -- no Python, ctypes or torch between the caller and the
library.  The reference-side entry points it stands for: ADmin.A_gradA_taped (_autodiffmin.py:57-58),
ADmin.min_lbfgs_scipy (:72-95), Annealer.anneal (va_ode.py:474-490, 707-789).  Results against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_program(tmp_path):
    import va_oracle
    from varanneal_amd import _build, twin
    lib = os.path.join(ROOT, "varanneal_amd")
    assert os.path.exists(os.path.join(lib, "libvaranneal_amd.so")), _build
    exe = tmp_path / "va_client"
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "c_client", "va_client.c"), "-L", lib, "-lvaranneal_amd",
                           "-Wl,-rpath," + lib])
    B, D, N, nbeta, maxiter = 3, 20, 120, 4, 40
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, b, Y, Lidx)
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    rf, alpha = 1.5 ** 5, 1.5
    prob = tmp_path / "problem.bin"
    with open(prob, "wb") as f:
        f.write(np.array([B, D, N, len(Lidx), 1, nbeta, maxiter, 1], dtype="<i4").tobytes())
        f.write(np.array([twin.DT, 4.0, 4e-6, rf, alpha], dtype="<f8").tobytes())
        f.write(np.asarray(Lidx, dtype="<i4").tobytes())
        f.write(np.ascontiguousarray(Y, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(P[:, 0], dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(XP, dtype="<f8").tobytes())
    out = tmp_path / "results.bin"
    r = subprocess.run([str(exe), str(prob), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = open(out, "rb").read()
    nv = N * D + 1
    off = [0]

    def take(dtype, n):
        a = np.frombuffer(raw, dtype=dtype, count=n, offset=off[0]); off[0] += a.nbytes
        return a
    ek = int(take("<i4", 1)[0])
    A1 = take("<f8", 3 * B).reshape(3, B); g = take("<f8", B * nv).reshape(B, nv)
    A2 = take("<f8", 3 * B).reshape(3, B); X2 = take("<f8", B * nv).reshape(B, nv)
    nit = take("<i4", B); nfev = take("<i8", B); st = take("<i4", B)
    ame = take("<f8", B * nbeta * 3).reshape(B, nbeta, 3); pest = take("<f8", B * nbeta).reshape(B, nbeta)
    nit3 = take("<i4", B * nbeta).reshape(B, nbeta)
    assert off[0] == len(raw) and ek == 4
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxiter': maxiter, 'maxfun': 1000000}
    for b in range(B):
        opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc="trapezoid")
        Ao, meo, feo, go = opb.action_grad(XP[b], rf)
        assert abs(A1[0, b] - Ao) <= 1e-12 * abs(Ao) and abs(A1[1, b] - meo) <= 1e-12 * abs(Ao) and abs(A1[2, b] - feo) <= 1e-12 * abs(Ao)
        assert np.abs(g[b] - go).max() <= 1e-10 * np.abs(go).max()
        x, Am, s, n_it, n_f = opb.minimize_lbfgs(XP[b], rf, opts)
        assert (nit[b], nfev[b], st[b]) == (n_it, n_f, s)
        assert abs(A2[0, b] - Am) <= 1e-6 * abs(Am) and np.abs(X2[b] - x).max() <= 1e-6
        ro = opb.anneal(XP[b], alpha, np.arange(nbeta), opts)
        assert list(nit3[b][:2]) == list(ro["nit"][:2])
        assert np.all(np.abs(ame[b, :, 0] - ro["A"]) <= 1e-6 * np.abs(ro["A"]) + 1e-8)
        assert np.all(np.abs(pest[b] - ro["minpaths"][:, -1]) <= 1e-3)
