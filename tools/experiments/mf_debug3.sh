#!/bin/bash
# Matrix-pipe filter experiment: does the corruption survive a build without packed-f32 VALU instructions?
D=tools/experiments/mf_debug.py
for n in nopk; do
  L=build/librt_$n.so
  RT_HIP_LIBRARY=$L timeout -k 10 200 python3 $D dump gpurun_out/x_ref_$n.npz > gpurun_out/x_dump_$n.log 2>&1 || exit 1
  RT_HIP_PRIMARY_MFMA=1 RT_HIP_LIBRARY=$L timeout -k 10 300 python3 $D cap gpurun_out/x_ref_$n.npz 64 > gpurun_out/x_cmp_$n.log 2>&1 || exit 1
  echo "$n done: $(grep -c 'fb diff 0, seg diff 0' gpurun_out/x_cmp_$n.log) clean reps of 64"
done
