#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_kernels.py -q --no-header -p no:cacheprovider -x -k "two_segments" > gpurun_out/u_pytest.log 2>&1; grep -v amdgpu.ids gpurun_out/u_pytest.log | tail -8
timeout 300 python tools/cross_attn_ab.py 5 2>/dev/null | grep -v amdgpu.ids > gpurun_out/u_cross_ab.log; cat gpurun_out/u_cross_ab.log
