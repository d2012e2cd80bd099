#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4p_pytest_vae.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4p_pytest_vae.log
grep -v amdgpu.ids gpurun_out/r4p_pytest_vae.log | tail -8
CE_VAE_BENCH_OUT=gpurun_out/r4p_vae_bench.json timeout 300 python tools/vae_bench.py 720 1280 5 > gpurun_out/r4p_vae_bench.log 2>&1
grep -E "^encode|^decode|attention_1head" -A3 gpurun_out/r4p_vae_bench.log | head -30
