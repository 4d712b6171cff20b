"""Multi-GPU sharding of independent annealing runs.

The reference's only parallelism is an SGE array job: one OS process, one tape id
and one random seed per task, results through the filesystem
(examples/nnet_barimages/SGEcluster/submit_multiM.sh:14-30, qsub_command.sh:8).
Here: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm),
seeds block-partitioned over ranks, NO collective on the data path, and ONE
all-gather of the per-seed result table at the end (latency-bound: KBs).
"""
import numpy as np


def rank_world(group=None):
    """(rank, world) of the torch.distributed job this process belongs to, or (0, 1)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def local_device(devices=None, rank=0):
    """device ordinal of this rank: devices[rank % len(devices)], else LOCAL_RANK, else 0"""
    import os
    if devices:
        return int(devices[rank % len(devices)])
    return int(os.environ.get("LOCAL_RANK", "0"))


def seed_range(n_seeds, rank, world):
    """Static block partition: rank r of R gets [r*n/R, (r+1)*n/R)."""
    lo = (n_seeds * rank) // world
    hi = (n_seeds * (rank + 1)) // world
    return lo, hi


def gather_tables(local, n_seeds, group=None, device=None):
    """All-gather per-seed result tables with ONE collective.

    `local` maps name -> float/int array whose leading axis is this rank's seeds (in
    seed order).  Returns the same names with leading axis `n_seeds` on every rank.
    `device`: the GPU ordinal this rank's problem ran on (RCCL backend): the collective's buffers are
    allocated THERE, whatever torch's current device is -- two ranks staging on cuda:0 would make RCCL
    fail with a duplicate-GPU error or hang.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    names = sorted(local)
    lo, hi = seed_range(n_seeds, rank, world)
    nloc = hi - lo
    widths, shapes = [], []
    for n in names:
        a = np.asarray(local[n])
        if a.shape[0] != nloc:
            raise ValueError("%s has %d rows, this rank owns %d seeds" % (n, a.shape[0], nloc))
        shapes.append(a.shape[1:])
        widths.append(int(np.prod(a.shape[1:], dtype=np.int64)))
    maxloc = max(seed_range(n_seeds, r, world)[1] - seed_range(n_seeds, r, world)[0] for r in range(world))
    row = sum(widths)
    buf = np.zeros((maxloc, row), dtype=np.float64)
    col = 0
    for n, w in zip(names, widths):
        buf[:nloc, col:col + w] = np.asarray(local[n], dtype=np.float64).reshape(nloc, w)
        col += w
    use_cuda = dist.get_backend(group) == "nccl"
    send = torch.from_numpy(buf)
    if use_cuda:
        send = send.to(torch.device("cuda", torch.cuda.current_device() if device is None else int(device)))
    recv = torch.empty((world * maxloc, row), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)        # the single RCCL gather
    recv = recv.cpu().numpy().reshape(world, maxloc, row)
    out = {n: np.empty((n_seeds,) + s, dtype=np.asarray(local[n]).dtype) for n, s in zip(names, shapes)}
    for r in range(world):
        rlo, rhi = seed_range(n_seeds, r, world)
        col = 0
        for n, w, s in zip(names, widths, shapes):
            out[n][rlo:rhi] = recv[r, :rhi - rlo, col:col + w].reshape((rhi - rlo,) + s)
            col += w
    return out
