"""Per-wave timeline of the C3 evaluation kernel (profiling only).  Needs the diagnostic library
(python -m varanneal_amd._build --stamps; run with VARANNEAL_AMD_LIB=.../libvaranneal_amd_stamps.so):
every wave of k_eval4 records wall_clock64 (100 MHz) at
  0 start | 1 loads issued (wave 0 of a seed's last workgroup: tail done) | 2 image landed | 3 rows+scatter done
  4 gather done, stores issued | 5 partial sums in LDS | 6 past the workgroup barrier | 7 (wave 0) arrival returned"""
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
info = pb.info()
nwg = ((B * info["ntiles"] + 7) // 8) * 8   # (workgroups: 256 with three sub-tiles per wave)
pb.action_grad(XP, bench.RF_SCALE)
pb.eval_timed(bench.RF_SCALE, 20)
names = ["start", "issued", "landed", "rows", "gather", "sums", "barrier", "arrived"]
for rep in range(3):
    pb.eval_timed(bench.RF_SCALE, 1)
    raw = pb.debug_partials(nwg * 4 * 10).view(np.uint64).reshape(nwg, 4, 10)[:, :, :8].astype(np.int64)
    t0 = raw[:, :, 0].min()
    us = (raw - t0) * 1e-2
    w0 = us[:, 0, :]
    print("rep %d: %d workgroups" % (rep, nwg))
    for k in (0, 2, 3, 4, 5, 6):
        a = us[:, :, k]
        print("   %-8s all waves: min %5.2f  median %5.2f  p90 %5.2f  max %5.2f us" % (names[k], a.min(), np.median(a), np.percentile(a, 90), a.max()))
    # the tail wave (wave 0 of each seed's last tile) re-uses slots: 7 tail starts, 2 rows validated, 5 outputs issued, 1 all its stores acknowledged
    nt = info["ntiles"]
    tw = np.array([us[w, 0, :] for w in range(nwg) if False])
    import itertools
    per = (nwg + 7) >> 3
    tails = []
    for bid in range(nwg):
        w = (bid & 7) * per + (bid >> 3)
        if w < B * nt and w % nt == nt - 1:
            tails.append(us[bid, 0, :])
    tails = np.array(tails)
    for k, nm in ((3, "rows done"), (7, "tail starts"), (2, "rows valid"), (5, "outputs out"), (4, "gather done"), (1, "stores acked")):
        print("   tail wave %-12s: min %5.2f median %5.2f max %5.2f us" % (nm, tails[:, k].min(), np.median(tails[:, k]), tails[:, k].max()))
    d = np.diff(us[:, :, [0, 2, 3, 4, 5, 6]], axis=2)
    print("   phase medians: wait-for-image %.2f  rows %.2f  gather %.2f  sums %.2f  barrier %.2f" % tuple(np.median(d[:, :, i]) for i in range(5)))
    grp = (np.arange(nwg) >> 8) % 3
    for gi in range(3):
        m = grp == gi
        print("   blockIdx group %d: start %.2f landed %.2f rows %.2f gather %.2f barrier %.2f" % (
            gi, np.median(us[m, :, 0]), np.median(us[m, :, 2]), np.median(us[m, :, 3]), np.median(us[m, :, 4]), np.median(us[m, :, 6])))
pb.close()
