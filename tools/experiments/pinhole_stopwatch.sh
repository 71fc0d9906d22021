#!/bin/bash
# usage (GPU box): tools/experiments/pinhole_stopwatch.sh [c2|c4 width height spp]  -- the frame with the -DRT_DBG_TIME=n builds build/ab/t{n}.so:
# cycle-counter sums over the waves per region (the "segments" of a stopwatch build are that sum; regions in rt_pinhole.hip)
W=${1:-c2}; X=${2:-1920}; Y=${3:-1080}; S=${4:-64}
for n in 1 2 3 4 5 6 7 8 11 12 13 14 15; do
  echo -n "region $n: "
  RT_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/ab/t$n.so timeout -k 10 120 python3 tools/time_frame.py $W $X $Y $S 8 3 2>&1 | grep median | sed 's/.*last launch segments //; s/;.*//'
done
