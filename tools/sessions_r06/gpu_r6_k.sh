#!/bin/bash
# round 6, session k: the toy-width fp8 edit's measured distances (to replace its 0.2 bound), then the whole suite timed after the trims
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_fp8_gpu.py -q -m gpu -s -k "edit_end_to_end" 2>&1 | grep "fp8 edit\|passed\|failed"
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 -x ) > $O/r6k_pytest.log 2>&1; tail -20 $O/r6k_pytest.log
