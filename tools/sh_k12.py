"""usage: python3 tools/sh_k12.py  -- Simpson-Hermite at the C3 shape with runs of 4 (the chooser's), 8 and 12 rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for kw in ({}, {"disc": "SimpsonHermite", "N": 1001}, {"disc": "SimpsonHermite", "N": 1001, "tile_rows": 96},
           {"disc": "SimpsonHermite", "N": 1001, "tile_rows": 144}, {"tile_rows": 144}):
    r = bench.extra_variant(0, **kw)
    print("%-46s tile_rows %3d  kernel %d K=%2d  %7.2f us  frac %.3f" % (r["workload"], kw.get("tile_rows", 0), r["eval_kernel"], r["run_rows"],
                                                                      r["us_per_eval_launch"], r["frac"]), flush=True)
