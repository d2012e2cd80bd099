#!/bin/bash
# round 6, session n: VAE mid-block attention with two query blocks per wave and the key split
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python tools/attn1_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/r6n_attn1_ab.txt
timeout 900 python -m pytest tests/test_vae_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/r6n_pytest_vae.txt
timeout 600 python tools/vae_bench.py 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/r6n_vae_bench.txt
