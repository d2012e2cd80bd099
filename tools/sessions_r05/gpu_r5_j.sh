#!/bin/bash
# round 5, session j: the 16x16x32 attention body (csrc/ce_attn16.hip): parity test, then A/B against the production 32x32x16 body in one process
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -k "16x16x32" > gpurun_out/r5j_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5j_pytest.log )
tail -12 gpurun_out/r5j_pytest.log
L=chronoedit_amd/lib/libchronoedit_hip.so
timeout 600 python tools/attn_body_ab.py $L@0 $L@16 > gpurun_out/r5j_attn_body_ab.txt 2>&1; cat gpurun_out/r5j_attn_body_ab.txt
