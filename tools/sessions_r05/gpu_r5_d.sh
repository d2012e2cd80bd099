#!/bin/bash
# round 5, session d: what a desynchronised start would buy the MX fp8 GEMM (F8_ABLATE 6 / 7: the first round's workgroups skip part of their K range)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 600 python tools/gemm_mxfp8_ab.py $L/libce_f8base.so $L/libce_f8abl6.so $L/libce_f8abl7.so $L/libce_f8abl5.so > gpurun_out/r5d_gemm_mxfp8_desync_probe.txt 2>&1
cat gpurun_out/r5d_gemm_mxfp8_desync_probe.txt
