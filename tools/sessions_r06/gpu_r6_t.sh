#!/bin/bash
# round 6, session t: the encoder's 3 -> 96 stem conv on the slab kernel (Cin = 32 padded frames); VAE tests, pipeline tests, the VAE timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/r6t_pytest.txt
timeout 600 python tools/vae_bench.py 2>&1 | grep -v amdgpu.ids | grep "encode\|decode" | head -12 | tee $O/r6t_vae_bench.txt
