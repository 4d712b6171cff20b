"""Per-workgroup timeline of the C3 eval kernel (profiling only): VA_DEBUG_EVAL=16 makes every
workgroup of k_eval3 record wall_clock64 at start / after staging / after phase B / after the
stores are issued / after they are acknowledged, plus its HW_ID and XCC_ID."""
import os
import sys

import numpy as np

os.environ["VA_DEBUG_EVAL"] = "16"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from varanneal_amd import _capi, twin  # noqa: E402

D, N, B = 20, 1000, 64
Y, Lidx, XP, P = bench.make_inputs(D, N, B, 0)
pb = _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid")
info = pb.info()
nwg = ((B * info["ntiles"] + 7) // 8) * 8
pb.action_grad(XP, bench.RF_SCALE)
for rep in range(3):
    pb.eval_timed(bench.RF_SCALE, 1)
    raw = pb.debug_partials(nwg * 8).view(np.uint64).reshape(nwg, 8)
    t = raw[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    tick = 1e-2                                   # wall_clock64: 100 MHz -> 10 ns
    st, ld, rows, issued, done = [(t[:, k] - t0) * tick for k in range(5)]
    hw = raw[:, 5]; cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; xcc = raw[:, 6] & 0xF
    key = xcc * 1000 + se * 16 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print("rep %d: %d workgroups on %d distinct (xcc,se,cu); per-CU count min/max %d/%d" % (rep, nwg, len(uniq), cnt.min(), cnt.max()))
    for name, a in (("start", st), ("staged", ld), ("rows done", rows), ("stores issued", issued), ("stores acked", done)):
        print("   %-14s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us" % (name, a.min(), np.median(a), np.percentile(a, 90), a.max()))
    print("   phase medians: stage %.2f  rows %.2f  grad+issue %.2f  ack %.2f us" % (
        np.median(ld - st), np.median(rows - ld), np.median(issued - rows), np.median(done - issued)))
    grp = (np.arange(nwg) >> 8) % 3          # a CU holds blockIdx b, b+256, b+512: dispatch order
    for gi in range(3):
        m = grp == gi
        print("   blockIdx group %d: start %.2f  stage %.2f  rows %.2f  grad %.2f  end %.2f" % (
            gi, np.median(st[m]), np.median((ld - st)[m]), np.median((rows - ld)[m]), np.median((issued - rows)[m]), np.median(done[m])))
    if rep == 2:
        for k in uniq[:4]:
            print("   CU", k, "holds blockIdx", np.nonzero(key == k)[0])
pb.close()
