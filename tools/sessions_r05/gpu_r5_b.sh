#!/bin/bash
# round 5, session b: the LDS-DMA issue schedule of the MX fp8 GEMM (F8_DMA_SCHED 0 = round 4: all 16 pieces in groups 5..7; 1..4 spread them)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 600 python tools/gemm_mxfp8_ab.py $L/libce_f8sched1.so $L/libce_f8sched5.so $L/libce_f8sched6.so $L/libce_f8sched7.so $L/libce_f8sched8.so $L/libce_f8sched9.so > gpurun_out/r5b_gemm_mxfp8_sched_ab2.txt 2>&1
cat gpurun_out/r5b_gemm_mxfp8_sched_ab2.txt
