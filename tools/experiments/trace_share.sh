# usage (GPU box, repo root): tools/experiments/trace_share.sh STEP -- kernel trace of rank 0's share of a C4 frame dealt round-robin to STEP ranks
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_share -o s -- python3 $R/tools/time_frame.py c4 3840 2160 1024 8 3 $1 > $R/gpurun_out/trace_share.log 2>&1
cd $R
for f in $(find gpurun_out/trace_share -name "*.db"); do python3 tools/rocpd_summary.py $f; done | head -12
