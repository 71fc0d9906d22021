#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh TAG [c2|c4]
# bench line, rocprofv3 kernel-trace of the same command, then four PMC passes (SQ, FETCH_SIZE, WRITE_SIZE, TCC atomics), one run each.
set -e
TAG=$1
WL=${2:-c2}
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --workload $WL --no-cpu-baseline"
if [ "$WL" = c2 ]; then TR="--steps 5 --warmup 1"; else TR="--steps 2 --warmup 0"; fi
timeout -k 10 400 python3 bench.py --workload $WL > gpurun_out/bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $WL -- $B $TR > $R/gpurun_out/prof_$TAG.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_${TAG}_sq -o $WL -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_fetch -o $WL -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_write -o $WL -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_atomic -o $WL -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_${TAG}_atomic.log 2>&1 || true
cd $R
{
  echo "# command: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --no-cpu-baseline $TR   (MI355X); PMC passes: --steps 1 --warmup 0"
  grep '^{' gpurun_out/bench_$TAG.log
  for d in prof_$TAG pmc_${TAG}_sq pmc_${TAG}_fetch pmc_${TAG}_write pmc_${TAG}_atomic; do
    for f in $(find gpurun_out/$d -name "*.db"); do python3 tools/rocpd_summary.py $f; done
  done | grep -v "rocclr\|at::native"
} > gpurun_out/profile_$TAG.txt 2>&1
