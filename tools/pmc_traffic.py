"""usage: python tools/pmc_traffic.py <tag> <workload name>  -- turn the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc.sh (gpurun_out/pmc_<tag>/p3, p4) into profiles/pmc_traffic.json: HBM bytes per launch of the
evaluation kernel (FETCH_SIZE doubled: the gfx950 correction of MI355X_MICROARCH.md, HBM section),
stamped with the hash of the kernel sources it was measured on (bench.py drops it when they change)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag, workload = sys.argv[1], sys.argv[2]
acc, cnt, names = collections.defaultdict(float), collections.Counter(), set()
for p in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s" % tag, "p[34]", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        if "k_eval" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            names.add(r["Kernel_Name"].split("(")[0])
fetch_kb, write_kb = acc["FETCH_SIZE"] / cnt["FETCH_SIZE"], acc["WRITE_SIZE"] / cnt["WRITE_SIZE"]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
try:
    allrec = json.load(open(path))
except (OSError, ValueError):
    allrec = {}
allrec[workload] = {"hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
                    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb, "kernel": sorted(names),
                    "kernel_source_hash": bench.kernel_source_hash(),
                    "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc.sh); "
                           "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md"}
json.dump(allrec, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({workload: allrec[workload]}))       # (copy this into profiles/pmc_traffic.json)
