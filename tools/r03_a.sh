#!/bin/bash
# round-3 first GPU pass: per-class VALU issue rates, the GPU suite, then bench + profile of C2 and of the simulate call shape
set -e
R=$GRAFT_REPO_ROOT
timeout -k 10 120 ./build/valu_rate > gpurun_out/r03_valu_rate.txt 2>&1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03a_tests.log 2>&1
tools/profile_bench.sh r03a c2
tools/profile_bench.sh r03a_sim simulate
