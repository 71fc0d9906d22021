# usage (GPU box, repo root): tools/experiments/pmc_depth1.sh TAG [library] -- VALU / SALU / LDS instruction counts of a primary-only C2 frame
set -e
R=$GRAFT_REPO_ROOT
TAG=$1
[ -n "$2" ] && export RT_HIP_LIBRARY=$2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_d1_$TAG -o d1 -- python3 $R/tools/time_frame.py c2 1920 1080 64 1 2 > $R/gpurun_out/pmc_d1_$TAG.log 2>&1
cd $R
for f in $(find gpurun_out/pmc_d1_$TAG -name "*.db"); do python3 tools/rocpd_summary.py $f; done | grep "pinhole_kernel<" | grep "SQ_INSTS_VALU\|calls\|  [0-9]* *[0-9]* *[0-9]* *9" > gpurun_out/pmc_d1_$TAG.txt
