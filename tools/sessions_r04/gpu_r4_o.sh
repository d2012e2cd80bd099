#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_vae_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "head_conv or golden or 720p" > gpurun_out/r4o_pytest_vae.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4o_pytest_vae.log
grep -v amdgpu.ids gpurun_out/r4o_pytest_vae.log | tail -15
CE_VAE_BENCH_OUT=gpurun_out/r4o_vae_bench_head.json timeout 300 python tools/vae_bench.py 720 1280 5 > gpurun_out/r4o_vae_bench_head.log 2>&1
CE_VAE_HEAD_CONV=0 CE_VAE_BENCH_OUT=gpurun_out/r4o_vae_bench_nohead.json timeout 300 python tools/vae_bench.py 720 1280 5 > gpurun_out/r4o_vae_bench_nohead.log 2>&1
grep -E "encode|decode|head|96->8" gpurun_out/r4o_vae_bench_head.log | head -30
echo ---
grep -E "encode|decode|head|96->8" gpurun_out/r4o_vae_bench_nohead.log | head -30
