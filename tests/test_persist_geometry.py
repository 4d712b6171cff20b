"""CPU: slice geometry of the persistent per-seed ladder kernel (csrc/va_persist_geo.h, shared by the kernel and the host),
checked through a g++ build of tests/cpu_emul/persist_check.cpp: every slice holds at least two rows, Simpson-Hermite slices
are even, the LDS budget holds, a seed never gets more workgroups than allowed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DISC = {"euler": 0, "trapezoid": 1, "SimpsonHermite": 2, "forwardmap": 3}


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("pz") / "persist_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "varanneal_amd", "csrc"), "-o", path,
                           os.path.join(ROOT, "tests", "cpu_emul", "persist_check.cpp")])
    return path


def geo(exe, N, D, L=7, NP=1, NPest=1, m=10, disc="trapezoid", maxG=256, wantT=0):
    out = subprocess.run([exe] + [str(v) for v in (N, D, L, NP, NPest, m, DISC[disc], maxG, wantT)], capture_output=True, text=True)
    assert out.returncode == 0
    w = out.stdout.split()
    return None if w[0] == "NO" else (int(w[1]), int(w[2]), int(w[3]))


def test_baseline_configs(exe):
    assert geo(exe, 200, 20)[:2] == (7, 30)                         # C1: one seed, N = 200
    assert geo(exe, 1000, 20)[:2] == (36, 28)                       # C2: one seed, N = 1000
    assert geo(exe, 161, 20, L=8, disc="SimpsonHermite")[:2] == (6, 28)      # the shipped example
    assert geo(exe, 1000, 20, maxG=4) is None                       # C3: 64 seeds share 256 CUs -> 4 workgroups each: no
    assert geo(exe, 5000, 200, L=80) is None                        # C4's width never fits
    assert geo(exe, 200, 20, maxG=32)[:2] == (7, 30)                # 8 seeds of N = 200


@pytest.mark.parametrize("disc", ["trapezoid", "SimpsonHermite", "euler", "forwardmap"])
def test_invariants_over_many_shapes(exe, disc):
    n_ok = 0
    for D in (2, 4, 10, 20, 36, 64, 100):
        for N in (2, 3, 5, 17, 101, 161, 200, 999, 1001, 3000):
            if disc == "SimpsonHermite" and N % 2 == 0:
                continue
            for m in (3, 10, 17):
                g = geo(exe, N, D, L=max(1, D // 3), NP=3, NPest=2, m=m, disc=disc)
                if g is None:
                    continue
                G, T, lds = g
                n_ok += 1
                assert T >= 2 and G >= 1 and G <= 256 and lds <= 160 * 1024
                assert G * T >= N and (G - 1) * T < N and N - (G - 1) * T >= 2
                assert disc != "SimpsonHermite" or T % 2 == 0
                # the automatic choice is the largest admissible slice: one more row does not fit (or is not admissible)
                assert geo(exe, N, D, L=max(1, D // 3), NP=3, NPest=2, m=m, disc=disc, wantT=T) == g
    assert n_ok > 80


def test_requested_slices(exe):
    assert geo(exe, 200, 20, wantT=8)[:2] == (25, 8)
    assert geo(exe, 200, 20, wantT=1) is None                       # two rows at least
    assert geo(exe, 200, 20, wantT=150) is None                     # LDS
    assert geo(exe, 201, 20, disc="SimpsonHermite", wantT=7) is None       # Simpson-Hermite: even slices
    assert geo(exe, 201, 20, wantT=100) is None                     # the last slice would hold one row
    assert geo(exe, 200, 20, maxG=5, wantT=8) is None               # more workgroups than the seed may have
