#!/bin/bash
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 600 python tools/gemm_ab.py $L/libgemm_staged.so $L/libgemm_direct_ra1.so $L/libgemm_direct_ra3.so $L/libgemm_direct_ra5.so 2>&1 | grep -v amdgpu.ids | grep epi2 | tee gpurun_out/r4x_gemm_direct_ra.txt
