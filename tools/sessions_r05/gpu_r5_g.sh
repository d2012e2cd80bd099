#!/bin/bash
# round 5, session g: fp8 cross-attention (kernel-level contract test, whole-DiT contract test, error / time probe at the full width),
# VAE timings with the 96-wide tile and the launcher's tile cost model
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_mxfp8_gemm_gpu.py -q -x -k "two_segment or dit_forward_with_mxfp8 or fused or edit_end_to_end" > gpurun_out/r5g_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5g_pytest.log )
tail -15 gpurun_out/r5g_pytest.log
timeout 600 python tools/fp8_cross_probe.py > gpurun_out/r5g_fp8_cross_probe.txt 2>&1; cat gpurun_out/r5g_fp8_cross_probe.txt
CE_VAE_BENCH_OUT=gpurun_out/r5g_vae_bench_720p.json timeout 600 python tools/vae_bench.py 720 1280 5 > gpurun_out/r5g_vae_bench.log 2>&1; tail -45 gpurun_out/r5g_vae_bench.log | cut -c1-200
