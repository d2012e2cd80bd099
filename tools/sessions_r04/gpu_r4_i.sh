#!/bin/bash
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "attention" -x > gpurun_out/r4i_pytest_attn.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4i_pytest_attn.log
grep -v amdgpu.ids gpurun_out/r4i_pytest_attn.log | tail -4
timeout 900 python tools/attn_body_ab.py $L/libattn_nostagger.so@0 $L/libchronoedit_hip.so@0 $L/libattn_r3.so@0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4i_attn_sp_stagger_ab.txt
