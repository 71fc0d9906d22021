#!/bin/bash
# usage (GPU box, repo root): tools/ab_sim.sh ROUNDS lib1.so lib2.so ...  -- interleaved A/B of library builds on the simulate call shape (4.1 M and 16.6 M rays)
R=$GRAFT_REPO_ROOT
N=$1; shift
for r in $(seq 1 $N); do
  for lib in "$@"; do
    export RT_HIP_LIBRARY=$R/$lib
    echo "== $lib round $r"
    timeout -k 10 120 python3 tools/bench_simulate.py 2 8 || exit 1
    timeout -k 10 120 python3 tools/bench_simulate.py 8 4 || exit 1
  done
done 2>&1 | grep -v amdgpu.ids
