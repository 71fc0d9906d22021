#!/bin/bash
# usage (GPU box, repo root): tools/ab_c3.sh ROUNDS lib1.so lib2.so ...  -- interleaved A/B on the mesh workload (C3 frame at 64 spp, checksum printed)
set -e
N=$1; shift
for r in $(seq 1 $N); do
  for lib in "$@"; do
    echo -n "$lib: "
    RT_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python3 tools/time_frame.py c3 1920 1080 64 8 3 2>&1 | grep median
  done
done
