#!/bin/bash
# usage (GPU box, repo root): tools/pmc_simulate.sh TAG  -- SQ counters of the explicit-ray kernel under tools/bench_simulate.py
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmcs_${TAG}_a -o s -- python3 $R/tools/bench_simulate.py 2 4 > $R/gpurun_out/pmcs_${TAG}_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmcs_${TAG}_b -o s -- python3 $R/tools/bench_simulate.py 2 4 > $R/gpurun_out/pmcs_${TAG}_b.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT -d $R/gpurun_out/pmcs_${TAG}_c -o s -- python3 $R/tools/bench_simulate.py 2 4 > $R/gpurun_out/pmcs_${TAG}_c.log 2>&1 || true
cd $R
for d in pmcs_${TAG}_a pmcs_${TAG}_b pmcs_${TAG}_c; do for f in $(find gpurun_out/$d -name "*.db"); do python3 tools/rocpd_summary.py $f; done; done | grep "paths_kernel\|trace_kernel" > gpurun_out/pmcs_${TAG}.txt
rm -rf gpurun_out/pmcs_${TAG}_a gpurun_out/pmcs_${TAG}_b gpurun_out/pmcs_${TAG}_c
