"""Many random initial paths of the Lorenz-96 problem, sharded over the GPUs of one node.

The reference's counterpart is an SGE array job (examples/nnet_barimages/SGEcluster/
submit_multiM.sh:14-30): one OS process, one random seed and one set of output files per task.
Here: one process per GPU, seeds block-partitioned over the ranks, every rank anneals its seeds
as ONE batch on its device with no communication, and a single RCCL all-gather of the per-seed
result tables (actions, exit flags, estimated forcing) closes the run.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/Lorenz96_D20/Lorenz96_multi_gpu.py --seeds 512 --N 1000 --nbeta 30
(also runs as a plain `python ...` on one GPU)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from varanneal_amd import parallel, twin, va_ode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--D", type=int, default=20)
    ap.add_argument("--N", type=int, default=200)
    ap.add_argument("--nbeta", type=int, default=30)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    D, N = args.D, args.N
    t, Y, _, Lidx = twin.make_twin(D, N)                          # identical bytes on every rank
    lo, hi = parallel.seed_range(args.seeds, rank, world)
    X0 = np.empty((hi - lo, N, D)); P0 = np.empty((hi - lo, 1))
    for b, s in enumerate(range(lo, hi)):                         # RNG keyed by the GLOBAL seed index
        X0[b], P0[b] = twin.initial_guess(N, D, s)

    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(Y, t=t)
    t0 = time.time()
    a.anneal(X0, P0, 1.5, np.arange(args.nbeta), 4.0, 4e-6, Lidx, [0], dt_model=twin.DT, init_to_data=True,
             disc='trapezoid', method='L-BFGS-B',
             opt_args={'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000},
             device=local_rank, verbose=False)
    local = {"A": a.A_array, "flags": a.exitflags.astype(np.int32), "k": a.minpaths[:, :, -1], "nfev": a.nfev_array}
    res = parallel.gather_tables(local, args.seeds)               # the single RCCL gather
    dt = time.time() - t0
    if rank == 0:
        best = int(np.argmin(res["A"][:, -1]))
        print("%d seeds x %d ladder steps on %d GPU(s): %.2f s, %d action+gradient evaluations"
              % (args.seeds, args.nbeta, world, dt, int(res["nfev"].sum())))
        print("lowest final action %.6e (seed %d), forcing estimate k = %.4f (truth %.2f)"
              % (res["A"][best, -1], best, res["k"][best, -1], twin.K_TRUE))
        if args.out:
            np.savez(args.out, **res)
    a.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
