#!/bin/bash
# A/B of the matrix-pipe shared-origin filter (RT_HIP_PRIMARY_MFMA=1) against the VALU filter on one box: parity tests with the
# variant switched on, then interleaved bench.py runs.
set -e
TAG=$1
RT_HIP_PRIMARY_MFMA=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "render or projector or golden or full or c1 or tmp or csg or bench" > gpurun_out/mfma_parity_$TAG.log 2>&1
for k in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_valu_$k.log 2>&1
  RT_HIP_PRIMARY_MFMA=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${TAG}_mfma_$k.log 2>&1
done
