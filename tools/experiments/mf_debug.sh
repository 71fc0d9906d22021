#!/bin/bash
# Matrix-pipe filter experiment: where do the continuing rays of the MFMA variant go wrong?  (debug builds under build/)
D=tools/experiments/mf_debug.py
B1=build/librt_dbg_b1.so
AG=build/librt_dbg_b1_agent.so
MF=build/librt_mf.so
RT_HIP_LIBRARY=$B1 timeout -k 10 200 python3 $D dump gpurun_out/dbg_ref.npz > gpurun_out/dbg_dump.log 2>&1 &&
RT_HIP_LIBRARY=$B1 timeout -k 10 200 python3 $D cmp gpurun_out/dbg_ref.npz 4 > gpurun_out/dbg_cmp_valu.log 2>&1 &&
RT_HIP_PRIMARY_MFMA=1 RT_HIP_LIBRARY=$B1 timeout -k 10 200 python3 $D cmp gpurun_out/dbg_ref.npz 16 > gpurun_out/dbg_cmp_mf.log 2>&1 &&
RT_HIP_PRIMARY_MFMA=1 RT_HIP_LIBRARY=$AG timeout -k 10 200 python3 $D cmp gpurun_out/dbg_ref.npz 16 > gpurun_out/dbg_cmp_mf_agent.log 2>&1 &&
RT_HIP_LIBRARY=$MF timeout -k 10 200 python3 $D time 20 > gpurun_out/dbg_time_valu.log 2>&1 &&
RT_HIP_PRIMARY_MFMA=1 RT_HIP_LIBRARY=$MF timeout -k 10 200 python3 $D time 20 > gpurun_out/dbg_time_mf.log 2>&1
echo "exit $?"
tail -3 gpurun_out/dbg_time_valu.log gpurun_out/dbg_time_mf.log
