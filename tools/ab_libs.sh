#!/bin/bash
# usage (GPU box, repo root): tools/ab_libs.sh ROUNDS lib1.so lib2.so ...   -- interleaved A/B of library builds (RT_HIP_LIBRARY):
# C2 frame (median of 10), C4 at 128 spp, and the simulate call shape; checksums printed so a wrong build shows.
set -e
R=$GRAFT_REPO_ROOT
N=$1; shift
for r in $(seq 1 $N); do
  for lib in "$@"; do
    export RT_HIP_LIBRARY=$R/$lib
    echo "== $lib round $r"
    timeout -k 10 120 python3 tools/time_frame.py c2 1920 1080 64 8 10
    timeout -k 10 120 python3 tools/time_frame.py c4 3840 2160 128 8 3
    timeout -k 10 120 python3 tools/bench_simulate.py 2 5
  done
done
