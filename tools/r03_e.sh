#!/bin/bash
# GPU: the suite with the shipped library, then an interleaved A/B against build/ab/base.so on the Projector frames and both simulate shapes
set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_tests.log 2>&1 || { tail -30 gpurun_out/r03_tests.log; exit 1; }
tail -1 gpurun_out/r03_tests.log
for r in 1 2; do
  for lib in build/ab/base.so raytracing_amd/librt_hip.so; do
    echo "== $lib"
    export RT_HIP_LIBRARY=$GRAFT_REPO_ROOT/$lib
    (timeout -k 10 120 python3 tools/bench_simulate_c4.py && timeout -k 10 120 python3 tools/time_frame.py c4 3840 2160 128 8 3 &&
     timeout -k 10 120 python3 tools/time_frame.py c2 1920 1080 64 8 10 && timeout -k 10 120 python3 tools/bench_simulate.py 2 10) 2>&1 |
      grep -v amdgpu | sed "s/; last launch.*//; s/; algorithmic.*//" || exit 1
  done
done
