#!/bin/bash
# round 5, session p: what the fp8 GEMM epilogue is made of (F8_ABLATE 5 = none at all, 8 = no GELU / block maximum, 9 = no global stores, 10 = no LDS staging / barriers), cold weights
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 900 python tools/gemm_mxfp8_ab.py --cold $L/libce_epibase.so $L/libce_epi5.so $L/libce_epi8.so $L/libce_epi9.so $L/libce_epi10.so > gpurun_out/r5p_gemm_mxfp8_epilogue_ablate.txt 2>&1
cat gpurun_out/r5p_gemm_mxfp8_epilogue_ablate.txt
