import sys, os
sys.path.insert(0, os.getcwd())
import bench
for kw in ({"rf_vec": True}, {"nskip": 2, "N": 1001}, {"rm_vec": True}):
    for tr in (0, 48, 60, 72, 96):
        r = bench.extra_variant(0, tile_rows=tr, **kw)
        print("%-46s tile_rows=%3d K=%d %7.2f us" % (r["workload"], tr, r["run_rows"], r["us_per_eval_launch"]), flush=True)
for tr in (0, 48, 72, 96):
    r = bench.extra_variant(0, tile_rows=tr, disc="SimpsonHermite", N=1001)
    print("%-46s tile_rows=%3d K=%d %7.2f us" % (r["workload"], tr, r["run_rows"], r["us_per_eval_launch"]), flush=True)
