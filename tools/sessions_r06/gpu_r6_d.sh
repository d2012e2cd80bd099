#!/bin/bash
# round 6, session d: the register-direct epilogue of the MX fp8 GEMM - parity tests, then the A/B against the staged (F8_EPI_LDS=1) build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests/test_mxfp8_gemm_gpu.py tests/test_fp8_gpu.py -x -q -m gpu 2>&1 | tail -8
L=chronoedit_amd/lib
timeout 900 python tools/gemm_mxfp8_ab.py --cold $L/libce_f8epilds.so $L/libchronoedit_hip.so 2>&1 | grep -v amdgpu.ids | tee $O/r6d_gemm_mxfp8_epilogue_ab.txt
timeout 900 python tools/gemm_mxfp8_ab.py --cold --sustain $L/libce_f8epilds.so $L/libchronoedit_hip.so 2>&1 | grep -v amdgpu.ids | tee -a $O/r6d_gemm_mxfp8_epilogue_ab.txt
