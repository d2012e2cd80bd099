#!/bin/bash
# round 5, session n: the fp8 GEMM schedules under SUSTAINED load (300 launches per timing, cold weights): is the stand-alone gain a clock artefact?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 900 python tools/gemm_mxfp8_ab.py --cold --sustain $L/libce_sched0.so $L/libce_sched1.so $L/libce_sched5.so > gpurun_out/r5n_gemm_mxfp8_sched_sustained.txt 2>&1
cat gpurun_out/r5n_gemm_mxfp8_sched_sustained.txt
