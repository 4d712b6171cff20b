"""CPU, world_size 2 (gloo): seed sharding + the single all-gather of per-seed result
tables that closes a multi-GPU run (varanneal_amd/parallel.py).  On the GPU box the
same code runs with backend "nccl" (RCCL) on device tensors (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_seeds, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from varanneal_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.seed_range(n_seeds, rank, world)
    seeds = np.arange(lo, hi)
    local = {"A": 0.5 + seeds[:, None] * np.ones((1, 4)),                 # (B_loc, nbeta)
             "pest": np.stack([seeds * 10.0, seeds * 10.0 + 1], 1)[:, :, None] * np.ones((1, 1, 1)),
             "status": (seeds % 3).astype(np.int32)}
    try:
        out = parallel.gather_tables(local, n_seeds)
        q.put((rank, {k: v.copy() for k, v in out.items()}))
    except Exception as e:                                  # surface the failure instead of a timeout
        q.put((rank, {"error": repr(e)}))
        raise
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_seeds", [8, 7])          # even split and ragged split
def test_gather_tables_world2(n_seeds):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_seeds, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seeds = np.arange(n_seeds)
    for r in range(world):
        o = got[r]
        assert "error" not in o, o
        assert o["A"].shape == (n_seeds, 4) and np.array_equal(o["A"][:, 2], 0.5 + seeds)
        assert o["pest"].shape == (n_seeds, 2, 1) and np.array_equal(o["pest"][:, 1, 0], seeds * 10.0 + 1)
        assert o["status"].dtype == np.int32 and np.array_equal(o["status"], seeds % 3)


def test_seed_range_partitions_exactly():
    from varanneal_amd.parallel import seed_range
    for n in (1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            cuts = [seed_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


# ---- the same through the drop-in: Annealer.anneal(..., n_seeds=) on 1 and on 2 ranks -------------------
def _anneal_seeds(n_seeds):
    """n_seeds short ladders through va_ode.Annealer with the oracle-backed stand-in device (as in
    tests/test_annealer_host.py); returns the per-seed tables `gathered` holds."""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import test_annealer_host as tah
    from varanneal_amd import _capi, twin, va_ode
    real, _capi.Problem = _capi.Problem, tah.OracleBackedProblem
    try:
        return _anneal_seeds_on_stand_in(n_seeds, tah, twin, va_ode)
    finally:
        _capi.Problem = real


def _anneal_seeds_on_stand_in(n_seeds, tah, twin, va_ode):
    D, N = 20, 60
    t, Y, _, Lidx = twin.make_twin(D, N)
    X0 = np.empty((n_seeds, N, D)); P0 = np.empty((n_seeds, 1))
    for s in range(n_seeds):                                # keyed by the GLOBAL seed index
        x, p = twin.initial_guess(N, D, s, Y, Lidx)
        X0[s] = x; P0[s] = p
    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(Y, t=t)
    a.anneal(X0, P0, 1.5, np.arange(0, 12, 3), 4.0, 4e-6, list(Lidx), [0], disc="trapezoid",
             opt_args=dict(tah.OPTS, maxiter=8), verbose=False, n_seeds=n_seeds)
    return a


def _anneal_worker(rank, world, port, n_seeds, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a = _anneal_seeds(n_seeds)
        q.put((rank, dict(a.gathered, lo=a.seed_lo, hi=a.seed_hi)))
    except Exception as e:
        q.put((rank, {"error": repr(e)}))
        raise
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_seeds", [4, 5, 1])         # even split, ragged split, more ranks than seeds
def test_annealer_seed_sharding_is_independent_of_world_size(n_seeds):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_anneal_worker, args=(r, world, port, n_seeds, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    one = _anneal_seeds(n_seeds)                           # the same seeds in ONE process
    assert (one.seed_lo, one.seed_hi) == (0, n_seeds)
    for r in range(world):
        o = got[r]
        assert "error" not in o, o
        for k in ("A", "me", "fe", "params", "exitflags", "nit", "nfev"):
            assert o[k].shape == one.gathered[k].shape and np.array_equal(o[k], one.gathered[k]), (r, k)
    assert got[0]["hi"] == got[1]["lo"] and got[1]["hi"] == n_seeds
