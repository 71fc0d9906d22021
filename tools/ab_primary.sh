#!/bin/bash
# A/B of two library builds on the C2 frame: depth 1 (primary segments only) and depth 8, kernel times from rocprofv3.
# "old" = a build of another revision copied to raytracing_amd/librt_hip_old.so (selected through RT_HIP_LIBRARY), "new" = the tree's.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in old new; do
  for depth in 1 8; do
    if [ $lib = old ]; then export RT_HIP_LIBRARY=$R/raytracing_amd/librt_hip_old.so; else unset RT_HIP_LIBRARY; fi
    timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ab_${lib}_$depth -o c2 -- python3 $R/tools/prof_render.py 1920 1080 64 3 c2 $depth > $R/gpurun_out/ab_${lib}_$depth.log 2>&1
    echo "$lib depth $depth: $(grep 'rep 2' $R/gpurun_out/ab_${lib}_$depth.log | cut -d: -f2 | cut -d, -f1) ; $(python3 $R/tools/rocpd_summary.py $R/gpurun_out/ab_${lib}_$depth/c2_results.db | grep pinhole | awk '{print $(NF-1)}') ns"
  done
done
