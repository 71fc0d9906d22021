#!/bin/bash
# usage: tools/c3_profile.sh TAG  — GPU tests, then kernel-trace and one SQ counter pass of the C3 render (32 spp)
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/parity_$TAG.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3$TAG -o c3 -- python3 $R/tools/prof_render.py 1920 1080 32 2 c3 > $R/gpurun_out/c3_prof_$TAG.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/pmc_c3$TAG -o c3 -- python3 $R/tools/prof_render.py 1920 1080 32 1 c3 > $R/gpurun_out/pmc_c3$TAG.log 2>&1
cd $R
for f in $(find gpurun_out/prof_c3$TAG gpurun_out/pmc_c3$TAG -name "*.db"); do python3 tools/rocpd_summary.py $f; done | grep -v "rocclr\|finalize" > gpurun_out/c3_$TAG.txt 2>&1
