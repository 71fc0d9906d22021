#!/bin/bash
# usage (GPU box, repo root): tools/ab_paths.sh ROUNDS lib1.so lib2.so ...  -- interleaved A/B of library builds on what the explicit-ray kernel serves:
# the simulate call shape (4.1 M and 16.6 M rays), the C4 primary rays as a list on the 10,000-sphere scene, a ParallelGrid render of the C2 cloud
set -e
R=$GRAFT_REPO_ROOT
N=$1; shift
for r in $(seq 1 $N); do
  for lib in "$@"; do
    export RT_HIP_LIBRARY=$R/$lib
    echo "== $lib round $r"
    timeout -k 10 120 python3 tools/bench_simulate.py 2 10
    timeout -k 10 120 python3 tools/bench_simulate.py 8 5
    timeout -k 10 120 python3 tools/bench_simulate_c4.py
    timeout -k 10 120 python3 tools/time_parallel_grid.py
  done
done
