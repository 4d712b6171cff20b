#!/bin/bash
# usage: tools/nn_pmc.sh <outdir>  -- HBM bytes of the c5x network kernels: FETCH_SIZE and WRITE_SIZE in separate passes
# (rocprofv3 --kernel-trace --pmc only; MI355X_MICROARCH.md: FETCH_SIZE counts 32-byte halves on gfx950 -> x 2)
out=$1; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p1 -- python3 tools/nn_c5x.py 20 > $out/p1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p2 -- python3 tools/nn_c5x.py 20 > $out/p2.log 2>&1 || exit 1
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
tot = 0.0
for k, v in sorted(acc.items()):
    if "nnet" not in k: continue
    f = 2.0 * v.get("FETCH_SIZE", 0.0) / max(cnt[(k, "FETCH_SIZE")], 1) * 1024 / 1e6
    w = v.get("WRITE_SIZE", 0.0) / max(cnt[(k, "WRITE_SIZE")], 1) * 1024 / 1e6
    tot += f + w
    print("%-42s read %8.1f MB  write %8.1f MB per launch" % (k, f, w))
print("all network kernels: %.1f MB per evaluation" % tot)
PY
