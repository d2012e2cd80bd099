#!/bin/bash
# round 6, session b: (1) the new build (hidden visibility, product + diagnostic library) on the GPU: selected tests; (2) configs[2] shapes (M = 7 200):
# 256- vs 384-row macro tile with the 64 MiB and a 160 MiB split-K scratch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_fp8_gpu.py tests/test_dit_forward_gpu.py -x -q -m gpu 2>&1 | tail -5
for ws in 64 160; do
  echo "== M = 7200, split-K scratch $ws MiB" >> $O/r6b_gemm_m7200.txt
  CE_GEMM_AB_M=7200 CE_GEMM_WS_MB=$ws timeout 600 python tools/gemm_variants.py 4,6 5 2>&1 | grep -v amdgpu.ids >> $O/r6b_gemm_m7200.txt
done
cat $O/r6b_gemm_m7200.txt
