#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -s -k "mxfp8" > gpurun_out/pytest_mxfp8.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_mxfp8.log
grep -E "passed|failed|mxfp8|DiT|Error|assert" gpurun_out/pytest_mxfp8.log | tail -14
