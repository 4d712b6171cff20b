#!/bin/bash
# round 4, call 1: bench line at the driver's flags and at the defaults; the ex1 ladder under rocprofv3
set -o pipefail
O=gpurun_out/r4a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err || exit 11
python bench.py --steps 2000 --warmup 200 --no-extra > $O/bench_2000.json 2> $O/bench_2000.err || exit 12
python tools/ex1_probe.py --nbeta 101 --graph 1 > $O/probe_plain.log 2>&1 || exit 13
prof() {   # tag, limit, probe arguments; a run killed at its limit ends the call (no GPU step after a timeout)
    tag=$1; lim=$2; shift 2
    timeout -k 10 $lim rocprofv3 --kernel-trace --stats -d $O/$tag -o $tag -- python3 tools/ex1_probe.py "$@" > $O/probe_$tag.log 2>&1
    rc=$?; echo "$tag rc=$rc" >> $O/rc.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then cat $O/rc.log; exit 14; fi
}
prof p30 200 --nbeta 30 --graph 1
prof p101g0 300 --nbeta 101 --graph 0
prof p101g1 300 --nbeta 101 --graph 1
find $O -name "*.csv" -size +2M -delete
cat $O/rc.log
