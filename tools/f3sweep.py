"""usage: python tools/f3sweep.py  -- the (f)3 variants of the evaluation at the C3 shape, one line each:
scalar weights / vector RF0 (D,) / vector RM / merr_nskip = 2 / Simpson-Hermite, against the scalar-weight trapezoid
kernel of the headline.  Reference: va_ode.py:147-148 (RM array), :203-209 (RF array), :404-437 (Simpson-Hermite),
:555-558 (merr_nskip)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

base = None
for kw in ({}, {"rf_vec": True}, {"rm_vec": True}, {"rf_vec": True, "rm_vec": True}, {"nskip": 2, "N": 1001},
           {"disc": "SimpsonHermite", "N": 1001}, {"disc": "SimpsonHermite", "N": 1001, "rf_vec": True}, {"disc": "euler"}, {"disc": "forwardmap"}):
    r = bench.extra_variant(0, **kw)
    if base is None:
        base = r["us_per_eval_launch"]
    print("%-52s kernel %d K=%d  %7.2f us  frac %.3f  x%.2f of the scalar-weight trapezoid" % (
        r["workload"], r["eval_kernel"], r["run_rows"], r["us_per_eval_launch"], r["frac"], r["us_per_eval_launch"] / base), flush=True)
