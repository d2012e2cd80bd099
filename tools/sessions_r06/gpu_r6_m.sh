#!/bin/bash
# round 6, session m: the tests changed after the last whole-suite run (sensitivity table inside the full-width block test, toy fp8 edit bound)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python tools/fp8_sensitivity.py 2>&1 | tee $O/r6m_fp8_sensitivity_from_test.txt
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_bench_shapes_gpu.py tests/test_ulysses.py -x -q -m gpu 2>&1 | tail -4
