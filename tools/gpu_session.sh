#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vae_gpu.py -q --no-header -p no:cacheprovider -x > gpurun_out/m_pytest_vae.log 2>&1; grep -v amdgpu.ids gpurun_out/m_pytest_vae.log | tail -15
CE_VAE_GEMM_CONV=0 timeout 300 python tools/vae_bench.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/m_vae_bench_old.log; tail -4 gpurun_out/m_vae_bench_old.log | cut -c1-1500
timeout 300 python tools/vae_bench.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/m_vae_bench_new.log; tail -4 gpurun_out/m_vae_bench_new.log | cut -c1-1500
