#!/bin/bash
# round 5, session c: one-ingredient-out builds of the MX fp8 GEMM main loop (F8_ABLATE 1..5) against the production build (schedule 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 600 python tools/gemm_mxfp8_ab.py $L/libce_f8base.so $L/libce_f8abl1.so $L/libce_f8abl2.so $L/libce_f8abl3.so $L/libce_f8abl4.so $L/libce_f8abl5.so > gpurun_out/r5c_gemm_mxfp8_ablate.txt 2>&1
cat gpurun_out/r5c_gemm_mxfp8_ablate.txt
