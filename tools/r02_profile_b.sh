#!/bin/bash
# usage (GPU box, repo root): tools/r02_profile_b.sh TAG
# A/B of the per-cell flush on one box (librt_hip_ab.so = a build with one atomic per lane and word in place of flush_cells, made by hand for this measurement), then the
# full profile of the shipped build (tools/profile_bench.sh: bench line, kernel trace, SQ / FETCH_SIZE / WRITE_SIZE passes)
# and the WRITE_SIZE pass of the A/B build.
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
for k in 1 2; do
  RT_HIP_LIBRARY=$R/raytracing_amd/librt_hip_ab.so timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_simple_$k.log 2>&1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_seg_$k.log 2>&1
done
bash tools/profile_bench.sh $TAG
cd /tmp && export TMPDIR=/tmp
RT_HIP_LIBRARY=$R/raytracing_amd/librt_hip_ab.so timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_write_simple -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_write_simple.log 2>&1
cd $R
for f in $(find gpurun_out/pmc_${TAG}_write_simple -name "*.db"); do python3 tools/rocpd_summary.py $f; done | grep -v "rocclr\|at::native" > gpurun_out/profile_${TAG}_write_simple.txt
