"""Where the stragglers of the C3 evaluation run (profiling only; stamps build: python -m varanneal_amd._build --stamps).
Per workgroup of k_eval4: start / image landed / rows done / gather done / past the barrier (wall_clock64, 100 MHz) with the
XCD, shader engine, CU and wave slot it ran on, its dispatch rank on that CU, its seed and tile."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VARANNEAL_AMD_LIB", os.path.join(ROOT, "varanneal_amd", "libvaranneal_amd_stamps.so"))
import bench  # noqa: E402
from varanneal_amd import _capi, twin  # noqa: E402

D, N, B = 20, 1000, 64
Y, Lidx, XP, P = bench.make_inputs(D, N, B, 0)
pb = _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid")
nt = pb.info()["ntiles"]
nwg = ((B * nt + 7) // 8) * 8
pb.action_grad(XP, bench.RF_SCALE)
pb.eval_timed(bench.RF_SCALE, 200)
for rep in range(4):
    pb.eval_timed(bench.RF_SCALE, int(os.environ.get("TL_ITERS", "1")))      # (stamps of the LAST launch remain)
    raw = pb.debug_partials(nwg * 4 * 10).view(np.uint64).reshape(nwg, 4, 10)
    ts = raw[:, :, :8].astype(np.int64)
    us = (ts - ts[:, :, 0].min()) * 1e-2
    hw = raw[:, 0, 8].astype(np.int64)
    xcc = (raw[:, 0, 9] >> np.uint64(32)).astype(np.int64) & 15
    work = (raw[:, 0, 9] & np.uint64(0xffffffff)).astype(np.int64)
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    seed, tile = work // nt, work % nt
    start = us[:, :, 0].min(axis=1); landed = us[:, :, 2].max(axis=1); rows = us[:, :, 3].max(axis=1)
    gather = us[:, :, 4].max(axis=1); barrier = us[:, :, 6].max(axis=1)
    # dispatch rank on its CU (by start time)
    rank = np.zeros(nwg, int)
    for c in np.unique(cuid):
        m = np.where(cuid == c)[0]
        rank[m[np.argsort(start[m])]] = np.arange(len(m))
    print("rep %d: %d workgroups on %d distinct CUs (%s per CU); rows done: median %.2f p90 %.2f max %.2f us" % (
        rep, nwg, len(np.unique(cuid)), np.bincount(np.bincount(cuid.astype(int))[np.unique(cuid)]).tolist(),
        np.median(rows), np.percentile(rows, 90), rows.max()))
    for name, key in (("XCD", xcc), ("dispatch rank on CU", rank), ("tile of the seed", tile), ("shader engine", se)):
        print("   by %s:" % name)
        for v in np.unique(key):
            m = key == v
            print("      %-3d n=%3d  start %.2f  landed %.2f  rows done median %.2f max %.2f  gather %.2f  barrier %.2f" % (
                v, m.sum(), np.median(start[m]), np.median(landed[m]), np.median(rows[m]), rows[m].max(), np.median(gather[m]), np.median(barrier[m])))
    # per seed: its slowest workgroup against its median one, and where that workgroup sat
    worst = []
    for b in range(B):
        m = np.where(seed == b)[0]
        i = m[np.argmax(rows[m])]
        worst.append((rows[i] - np.median(rows[m]), rank[i], tile[i], xcc[i], landed[i] - np.median(landed[m]), start[i] - np.median(start[m])))
    worst = np.array(worst)
    print("   per seed, slowest workgroup minus the seed's median: rows done +%.2f us (max +%.2f); of that, image landed +%.2f, started +%.2f" % (
        np.median(worst[:, 0]), worst[:, 0].max(), np.median(worst[:, 4]), np.median(worst[:, 5])))
    print("   the slowest workgroup's dispatch rank on its CU: %s; its tile: %s; XCDs spanned by a seed's tiles: %s" % (
        np.bincount(worst[:, 1].astype(int), minlength=3).tolist(), np.bincount(worst[:, 2].astype(int), minlength=nt).tolist(),
        np.bincount([len(np.unique(xcc[seed == b])) for b in range(B)]).tolist()))
    # waves of one workgroup: spread of rows-done inside a workgroup
    print("   inside a workgroup: rows done of its slowest wave minus its fastest: median %.2f max %.2f us" % (
        np.median(us[:, :, 3].max(axis=1) - us[:, :, 3].min(axis=1)), (us[:, :, 3].max(axis=1) - us[:, :, 3].min(axis=1)).max()))
pb.close()
