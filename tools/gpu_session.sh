#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vae_gpu.py -q --no-header -p no:cacheprovider -x > gpurun_out/n_pytest_vae.log 2>&1; grep -v amdgpu.ids gpurun_out/n_pytest_vae.log | tail -15
timeout 300 python tools/conv_gemm_ab.py 3 2>/dev/null | grep -v amdgpu.ids > gpurun_out/n_conv_gemm_ab.log; cat gpurun_out/n_conv_gemm_ab.log
timeout 300 python tools/vae_bench.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/n_vae_bench_new.log; grep -E "encode|decode \(" gpurun_out/n_vae_bench_new.log
