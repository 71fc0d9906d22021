#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh TAG
# bench line, rocprofv3 kernel-trace of the same command, then three PMC passes (SQ, FETCH_SIZE, WRITE_SIZE), one run each.
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > gpurun_out/bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o c2 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_${TAG}_sq -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_fetch -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_write -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
cd $R
{
  echo "# command: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline   (C2, MI355X); PMC passes: --steps 1 --warmup 0"
  grep '^{' gpurun_out/prof_$TAG.log
  for d in prof_$TAG pmc_${TAG}_sq pmc_${TAG}_fetch pmc_${TAG}_write; do
    for f in $(find gpurun_out/$d -name "*.db"); do python3 tools/rocpd_summary.py $f; done
  done | grep -v "rocclr\|at::native"
} > gpurun_out/profile_$TAG.txt 2>&1
