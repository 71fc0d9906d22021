# usage (GPU box, repo root): tools/experiments/pmc_share.sh STEP -- VALU instruction count of rank 0's share of a C4 frame dealt round-robin to STEP ranks
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/pmc_share$1 -o s -- python3 $R/tools/time_frame.py c4 3840 2160 1024 8 2 $1 > $R/gpurun_out/pmc_share$1.log 2>&1
cd $R
for f in $(find gpurun_out/pmc_share$1 -name "*.db"); do python3 tools/rocpd_summary.py $f; done | grep "pinhole_kernel<" | grep "SQ_\|TraceParams)  "
