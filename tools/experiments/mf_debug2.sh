#!/bin/bash
# Matrix-pipe filter experiment: which input of the primary segment's shading differs when a continuing ray goes wrong?
# Needs build/librt_cap$n.so = the library with tools/experiments/mfma_primary_filter.patch applied and -DRT_DBG_CAP=$n.
D=tools/experiments/mf_debug.py
for n in 1 2 3 4 5 6 7 8 9 10 11; do
  L=build/librt_cap$n.so
  RT_HIP_LIBRARY=$L timeout -k 10 200 python3 $D dump gpurun_out/cap_ref$n.npz > gpurun_out/cap_dump$n.log 2>&1 || exit 1
  RT_HIP_PRIMARY_MFMA=1 RT_HIP_LIBRARY=$L timeout -k 10 200 python3 $D cap gpurun_out/cap_ref$n.npz 24 > gpurun_out/cap_cmp$n.log 2>&1 || exit 1
  echo "cap $n done"
done
