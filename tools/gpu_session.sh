#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vae_gpu.py -q --no-header -p no:cacheprovider -x > gpurun_out/q_pytest_vae.log 2>&1; grep -v amdgpu.ids gpurun_out/q_pytest_vae.log | tail -15
timeout 300 python tools/vae_bench.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/q_vae_bench.log; grep -E "encode|decode \(|replay" gpurun_out/q_vae_bench.log
R=$GRAFT_REPO_ROOT
cd /tmp
CE_VAE_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_vae_prof -o p -- python $R/tools/vae_bench.py > $R/gpurun_out/q_vae_prof.log 2>&1
cd $R
head -14 gpurun_out/q_vae_prof/p_kernel_stats.csv | cut -c1-200
