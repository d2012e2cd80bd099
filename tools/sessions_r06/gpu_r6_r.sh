#!/bin/bash
# round 6, session r: the slab kernel of the 96-channel convs (conv3x3_c96_kernel): parity, A/B against the GEMM tiles, the VAE
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_vae_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/r6r_pytest_vae.txt
CE_CONV_AB_ONLY96=1 timeout 600 python tools/conv_gemm_ab.py 3 2>&1 | grep -v amdgpu.ids | tee $O/r6r_conv_ab.txt
timeout 600 python tools/vae_bench.py 2>&1 | grep -v amdgpu.ids | grep "encode\|decode" | head -12 | tee $O/r6r_vae_bench.txt
