#!/bin/bash
# round-4 GPU session B: attention bodies (parity + A/B), then the test files session A did not reach
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "attention" -x > gpurun_out/r4b_pytest_attn.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4b_pytest_attn.log
grep -v amdgpu.ids gpurun_out/r4b_pytest_attn.log | tail -25
timeout 600 python tools/attn_body_ab.py chronoedit_amd/lib/libattn_r3.so@0 chronoedit_amd/lib/libchronoedit_hip.so@0 chronoedit_amd/lib/libchronoedit_hip.so@128 chronoedit_amd/lib/libchronoedit_hip.so@129 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4b_attn_body_ab.txt
timeout 600 python tools/cross_attn_ab.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4b_cross_attn_ab.txt
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_dit_forward_gpu.py tests/test_pipeline_gpu.py tests/test_bench_shapes_gpu.py tests/test_run_inference_main_gpu.py tests/test_ref_loop_gpu.py tests/test_adapters_gpu.py tests/test_fp8_gpu.py tests/test_ulysses.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r4b_pytest_rest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4b_pytest_rest.log
grep -v amdgpu.ids gpurun_out/r4b_pytest_rest.log | tail -15
