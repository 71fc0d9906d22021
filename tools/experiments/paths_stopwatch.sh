#!/bin/bash
# usage (GPU box): tools/experiments/paths_stopwatch.sh  -- the simulate call shape with the -DRT_DBG_PTIME=1..9 builds build/ab/pt{n}.so:
# cycles summed over the waves per region (the "segments" of a stopwatch build are that sum)
for n in 1 2 3 4 5 6 7 8 9; do
  echo -n "region $n: "
  RT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/ab/pt$n.so timeout -k 10 120 python3 tools/bench_simulate.py 2 3 2>&1 | grep "rays," | sed 's/ per call.*//'
done
