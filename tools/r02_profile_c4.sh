#!/bin/bash
# usage (GPU box, repo root): tools/r02_profile_c4.sh TAG — C4's streamed kernel (pinhole_kernel<2,true,false> + pinhole_table_kernel):
# kernel trace and SQ / FETCH_SIZE / WRITE_SIZE passes of `tools/prof_render.py 3840 2160 8 3 c4` (full grid and scene, 8 spp).
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
C4="python3 $R/tools/prof_render.py 3840 2160 8 3 c4"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_c4 -o c4 -- $C4 > $R/gpurun_out/prof_${TAG}_c4.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/pmc_${TAG}_c4_sq -o c4 -- $C4 > $R/gpurun_out/pmc_${TAG}_c4_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_c4_fetch -o c4 -- $C4 > $R/gpurun_out/pmc_${TAG}_c4_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_c4_write -o c4 -- $C4 > $R/gpurun_out/pmc_${TAG}_c4_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${TAG}_c4_sq2 -o c4 -- $C4 > $R/gpurun_out/pmc_${TAG}_c4_sq2.log 2>&1
cd $R
{
  echo "# command: rocprofv3 --kernel-trace --stats -- python3 tools/prof_render.py 3840 2160 8 3 c4   (C4 scene: 10,000 spheres, 3840x2160 cells, 8 spp, depth 8; MI355X); PMC passes: same command"
  grep "^rep\|^scene" gpurun_out/prof_${TAG}_c4.log
  for d in prof_${TAG}_c4 pmc_${TAG}_c4_sq pmc_${TAG}_c4_fetch pmc_${TAG}_c4_write pmc_${TAG}_c4_sq2; do
    for f in $(find gpurun_out/$d -name "*.db"); do python3 tools/rocpd_summary.py $f; done
  done | grep -v "rocclr\|at::native"
} > gpurun_out/profile_${TAG}_c4.txt 2>&1
