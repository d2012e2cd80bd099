#!/bin/bash
# round-4 GPU session D: w4 attention body v2 (cross-phase prefetch, mid-phase barrier, NSB = 1 form): parity + A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "attention_vt" -x > gpurun_out/r4d_pytest_attn.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4d_pytest_attn.log
grep -v amdgpu.ids gpurun_out/r4d_pytest_attn.log | tail -12
timeout 600 python tools/attn_body_ab.py chronoedit_amd/lib/libchronoedit_hip.so@0 chronoedit_amd/lib/libchronoedit_hip.so@128 chronoedit_amd/lib/libchronoedit_hip.so@129 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4d_attn_body_ab.txt
