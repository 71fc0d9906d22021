#!/bin/bash
# usage (GPU box): tools/experiments/walk_counters.sh PREFIX...  -- C2 frame with the -DRT_DBG_WALK=1 / 2 / 4 builds build/ab/PREFIX_w{1,2,4}.so:
# wave-passes, lane-passes and walks per frame (the segment total of a counter build is the counter)
for p in "$@"; do
  for n in 1 2 4; do
    echo -n "$p RT_DBG_WALK=$n: "
    RT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/ab/${p}_w$n.so timeout -k 10 120 python3 tools/time_frame.py c2 1920 1080 64 8 3 2>&1 | grep median | sed 's/.*last launch//'
  done
done
