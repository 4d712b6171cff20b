#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <bench args...>  -- memory-side request counters of the evaluation kernel (own runs,
# --kernel-trace only): sizes of the L2 -> fabric read / write requests and the write-queue stalls
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcm_$tag
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcm_$tag/p$n -- python3 bench.py --steps 30 --warmup 5 --no-cpu --no-extra $BENCH_ARGS > gpurun_out/pmcm_$tag/p$n.log 2>&1; echo "pass $n rc=$?"; }
BENCH_ARGS="$*"
run 1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
run 2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run 3 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob("gpurun_out/pmcm_$tag/p*/*/*counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"][:44]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        if "eval" in k:
            print(p.split("/")[2], k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
PY
