"""usage: python3 tools/sh_rf.py  -- Simpson-Hermite with an RF0 array at the C3 shape, runs of 4 (the chooser's), 6 and 8 rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for kw in ({"disc": "SimpsonHermite", "N": 1001, "rf_vec": True}, {"disc": "SimpsonHermite", "N": 1001, "rf_vec": True, "tile_rows": 72},
           {"disc": "SimpsonHermite", "N": 1001, "rf_vec": True, "tile_rows": 96}, {"disc": "SimpsonHermite", "N": 1001, "rm_vec": True},
           {"disc": "SimpsonHermite", "N": 1001, "rm_vec": True, "tile_rows": 72}, {"disc": "SimpsonHermite", "N": 1001, "nskip": 2},
           {"disc": "SimpsonHermite", "N": 1001, "nskip": 2, "tile_rows": 72}, {"disc": "SimpsonHermite", "N": 1001, "nskip": 2, "tile_rows": 48}):
    r = bench.extra_variant(0, **kw)
    print("%-50s tile_rows %3d  kernel %d K=%2d  %7.2f us" % (r["workload"], kw.get("tile_rows", 0), r["eval_kernel"], r["run_rows"], r["us_per_eval_launch"]), flush=True)
