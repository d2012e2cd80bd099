#!/bin/bash
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 600 python tools/gemm_ab.py $L/libgemm_base.so $L/libgemm_noepi.so $L/libgemm_stag8000.so $L/libgemm_stag16000.so $L/libgemm_stag32000.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4x_gemm_stagger.txt
