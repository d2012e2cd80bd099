#!/bin/bash
# round-2 session H: attention with V^T by LDS-DMA - parity + A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention" > gpurun_out/pytest_h.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_h.log
tail -15 gpurun_out/pytest_h.log
timeout 300 python tools/microbench.py attn 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench_attn_vt.txt; cat gpurun_out/microbench_attn_vt.txt
