#!/bin/bash
# round 3, session E: the new boundary / multi-rank tests
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ulysses.py tests/test_run_inference_main_gpu.py tests/test_ref_loop_gpu.py tests/test_pipeline_gpu.py tests/test_vae_gpu.py tests/test_unipc.py -m gpu -q --no-header -p no:cacheprovider -s > gpurun_out/e_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/e_pytest.log
grep -v "amdgpu.ids\|Gloo\|socket.cpp" gpurun_out/e_pytest.log | tail -60
