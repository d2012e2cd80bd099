#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "attention_vt" -x > gpurun_out/r4g_pytest_attn.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4g_pytest_attn.log
grep -v amdgpu.ids gpurun_out/r4g_pytest_attn.log | tail -4
timeout 600 python tools/attn_body_ab.py chronoedit_amd/lib/libchronoedit_hip.so@0 chronoedit_amd/lib/libattn_w4v3.so@128 chronoedit_amd/lib/libchronoedit_hip.so@128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4g_attn_body_ab.txt
CE_ATTN_WAVES=128 bash tools/gpu_pmc.sh attnvt_28800_body128_v4 attnvt 28800 40 1 2 > gpurun_out/r4g_pmc_attnvt_28800_body128_v4.txt 2>&1
grep -A12 "attn_fwd" gpurun_out/r4g_pmc_attnvt_28800_body128_v4.txt | grep -v "^--" | awk '{$1=$1};1' | sort -u | grep -E "GRBM|WAIT|WAVE_CYC|ACTIVE_INST_ANY|MFMA_BUSY"
