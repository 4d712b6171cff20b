#!/bin/bash
# usage: tools/ablate.sh <workload> <tile_rows> -- eval-kernel ablations (profiling only)
for dbg in 0 8 2; do
  echo -n "dbg=$dbg  "
  VA_DEBUG_EVAL=$dbg ./tools/sweep.sh $1 1000 "$2" 3
done
