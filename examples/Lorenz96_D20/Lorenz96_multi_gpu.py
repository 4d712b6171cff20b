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
from varanneal_amd import twin, va_ode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--D", type=int, default=20)
    ap.add_argument("--N", type=int, default=200)
    ap.add_argument("--nbeta", type=int, default=30)
    ap.add_argument("--out", default=None)
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with one GPU: every rank uses cuda:0 and the gather runs over gloo "
                         "(RCCL refuses two ranks on one device)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.share_gpu:
        local_rank = 0
        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    D, N = args.D, args.N
    t, Y, _, Lidx = twin.make_twin(D, N)                          # identical bytes on every rank
    X0 = np.empty((args.seeds, N, D)); P0 = np.empty((args.seeds, 1))
    for s in range(args.seeds):                                   # RNG keyed by the GLOBAL seed index: every
        X0[s], P0[s] = twin.initial_guess(N, D, s)                # rank builds the same arrays, uses its block

    a = va_ode.Annealer()
    a.set_model(twin.l96, D)
    a.set_data(Y, t=t)
    t0 = time.time()
    # n_seeds=: this rank anneals seeds [rank*n/world, (rank+1)*n/world) on its GPU as one batch; the single
    # all-gather (RCCL) leaves every seed's tables on every rank in a.gathered
    a.anneal(X0, P0, 1.5, np.arange(args.nbeta), 4.0, 4e-6, Lidx, [0], dt_model=twin.DT, init_to_data=True,
             disc='trapezoid', method='L-BFGS-B',
             opt_args={'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000},
             verbose=False, n_seeds=args.seeds, devices=[local_rank])
    res = {"A": a.gathered["A"], "flags": a.gathered["exitflags"], "k": a.gathered["params"][:, :, 0],
           "nfev": a.gathered["nfev"]}
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
