#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/attn_quant_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4u_attn_quant_ab.txt
