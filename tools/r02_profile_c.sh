#!/bin/bash
# usage (GPU box, repo root): tools/r02_profile_c.sh TAG — new GPU test, bench lines with checksums (C2, C4 full), the N > 1
# code path rehearsed on one GPU (one-rank RCCL group; two ranks over gloo), and the L2 atomic / write request counters.
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -k "bench_sharded" -x -q --timeout 200 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1
timeout -k 10 300 python3 bench.py --print-checksum > gpurun_out/bench_${TAG}_c2.log 2>&1
timeout -k 10 300 python3 bench.py --workload c4 --steps 1 --warmup 0 --print-checksum > gpurun_out/bench_${TAG}_c4.log 2>&1
RT_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --workload c4 --spp 16 --steps 3 --warmup 1 --print-checksum > gpurun_out/bench_${TAG}_c4_rccl1.log 2>&1
RT_BENCH_SINGLE_DEVICE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --spp 16 --steps 3 --warmup 1 --print-checksum > gpurun_out/bench_${TAG}_c4_gloo2.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_ATOMIC_sum -d $R/gpurun_out/pmc_${TAG}_tcc -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_tcc.log 2>&1
RT_HIP_LIBRARY=$R/raytracing_amd/librt_hip_ab.so timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_ATOMIC_sum -d $R/gpurun_out/pmc_${TAG}_tcc_simple -o c2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_tcc_simple.log 2>&1
cd $R
for d in pmc_${TAG}_tcc pmc_${TAG}_tcc_simple; do for f in $(find gpurun_out/$d -name "*.db"); do python3 tools/rocpd_summary.py $f; done; done | grep -v "rocclr\|at::native" > gpurun_out/profile_${TAG}_tcc.txt
