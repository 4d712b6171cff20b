#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   -- rocprofv3 counter passes for the eval kernel (own runs, no tracing domains)
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$tag
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag/p$n -- python3 bench.py --steps 40 --warmup 10 --no-cpu --no-extra $BENCH_ARGS > gpurun_out/pmc_$tag/p$n.log 2>&1; echo "pass $n rc=$?"; }
BENCH_ARGS="$*"
run 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run 2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run 3 FETCH_SIZE
run 4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob("gpurun_out/pmc_$tag/p*/*/*counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,v in acc.items():
        if "eval" in k or "update" in k or "direction" in k:
            print(p.split("/")[2], k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
PY
