#!/bin/bash
# usage (GPU box, repo root): tools/pmc_frame.sh TAG depth   -- SQ counters of C2 frames at a given max depth
TAG=$1; DEPTH=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_${TAG} -o c2 -- python3 $R/tools/time_frame.py c2 1920 1080 64 $DEPTH 3 > $R/gpurun_out/pmc_${TAG}.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc_${TAG}_b -o c2 -- python3 $R/tools/time_frame.py c2 1920 1080 64 $DEPTH 3 > $R/gpurun_out/pmc_${TAG}_b.log 2>&1 || exit 1
cd $R
for d in pmc_${TAG} pmc_${TAG}_b; do for f in $(find gpurun_out/$d -name "*.db"); do python3 tools/rocpd_summary.py $f; done; done | grep "pinhole_kernel" > gpurun_out/pmc_${TAG}.txt
