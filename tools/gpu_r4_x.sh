#!/bin/bash
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 600 python tools/gemm_ab.py $L/libgemm_staged.so $L/libgemm_direct.so $L/libgemm_noepi.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4x_gemm_direct_epi.txt
