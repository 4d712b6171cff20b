import sys, json
sys.path.insert(0, '.')
import bench
for kw in (dict(D=200, N=5000), dict(D=200, N=5000, rf_vec=True), dict(D=200, N=5000, rf_vec=True, rm_vec=True), dict(D=200, disc="SimpsonHermite", N=5001),
           dict(D=200, N=4999, nskip=2)):
    r = bench.extra_variant(0, steps=30, **kw)
    print("%-55s kernel %d  %.1f us  frac %.3f" % (r["workload"], r["eval_kernel"], r["us_per_eval_launch"], r["frac"]), flush=True)
r = bench.extra_c4(0)
print("c4", r["us_per_eval_launch"], r["frac"])
