#!/bin/bash
# round 5, session m: the LDS-DMA schedules of the MX fp8 GEMM again, with COLD weights (every launch another copy of W: HBM, as in the step)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 900 python tools/gemm_mxfp8_ab.py --cold $L/libce_sched0.so $L/libce_sched1.so $L/libce_sched2.so $L/libce_sched5.so $L/libce_sched6.so $L/libce_sched8.so > gpurun_out/r5m_gemm_mxfp8_sched_cold.txt 2>&1
cat gpurun_out/r5m_gemm_mxfp8_sched_cold.txt
