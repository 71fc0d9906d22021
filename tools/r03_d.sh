#!/bin/bash
# GPU: the suite with the shipped library, then an interleaved A/B of the libraries named on the command line (default: build/ab/base.so and the shipped one)
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03_tests.log 2>&1 || { tail -30 gpurun_out/r03_tests.log; exit 1; }
tail -1 gpurun_out/r03_tests.log
LIBS="${@:-build/ab/base.so raytracing_amd/librt_hip.so}"
tools/ab_libs.sh 2 $LIBS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_d.txt | grep "==\|median\|rays" | sed 's/; last launch.*//; s/; algorithmic.*//'
