#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh TAG [c2|c4|c3|simulate]
# bench line, rocprofv3 kernel-trace of the same command, then PMC passes, one run each: SQ issue / lane counters, the VALU
# instruction classes (SQ_INSTS_VALU_*), executed FP32 FLOPs, FETCH_SIZE, WRITE_SIZE, TCC atomics.  The summary
# (gpurun_out/profile_TAG.txt) starts with the hash of the sources the library was built from (tools/csrc_hash.py);
# tools/pmc_constants.py turns it into the entry of profiles/pmc_constants.json that bench.py reads.
set -e
TAG=$1
WL=${2:-c2}
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --workload $WL --no-cpu-baseline"
case $WL in
  c2|simulate) TR="--steps 5 --warmup 1" ;;
  c3) TR="--steps 2 --warmup 1" ;;
  *) TR="--steps 2 --warmup 0" ;;
esac
timeout -k 10 400 python3 bench.py --workload $WL --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $WL -- $B $TR > $R/gpurun_out/prof_$TAG.log 2>&1
P="--steps 1 --warmup 0"
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_${TAG}_sq -o $WL -- $B $P > $R/gpurun_out/pmc_${TAG}_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $R/gpurun_out/pmc_${TAG}_mix -o $WL -- $B $P > $R/gpurun_out/pmc_${TAG}_mix.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_IOPS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH -d $R/gpurun_out/pmc_${TAG}_flops -o $WL -- $B $P > $R/gpurun_out/pmc_${TAG}_flops.log 2>&1 || true
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_fetch -o $WL -- $B $P > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_write -o $WL -- $B $P > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_atomic -o $WL -- $B $P > $R/gpurun_out/pmc_${TAG}_atomic.log 2>&1 || true
cd $R
{
  echo "# command: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --no-cpu-baseline $TR   (MI355X); PMC passes: $P"
  echo "# csrc_sha256: $(python3 tools/csrc_hash.py)"
  grep '^{' gpurun_out/bench_$TAG.log
  for d in prof_$TAG pmc_${TAG}_sq pmc_${TAG}_mix pmc_${TAG}_flops pmc_${TAG}_fetch pmc_${TAG}_write pmc_${TAG}_atomic; do
    for f in $(find gpurun_out/$d -name "*.db" 2>/dev/null); do python3 tools/rocpd_summary.py $f; done
  done | grep -v "rocclr\|at::native"
} > gpurun_out/profile_$TAG.txt 2>&1
# the rocpd databases are large (gpurun copies back at most 64 MiB): keep the text summary only
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_mix gpurun_out/pmc_${TAG}_flops gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_atomic
