#!/bin/bash
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 900 python tools/attn_body_ab.py $L/libchronoedit_hip.so@128 $L/libattn_abl1.so@128 $L/libattn_abl2.so@128 $L/libattn_abl4.so@128 $L/libattn_abl8.so@128 $L/libattn_abl16.so@128 $L/libattn_abl24.so@128 $L/libattn_abl32.so@128 $L/libattn_abl64.so@128 $L/libattn_abl96.so@128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4h_attn_w4_ablate.txt
