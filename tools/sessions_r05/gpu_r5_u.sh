#!/bin/bash
# round 5, session u: bench.py's own tests on the final bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_bench_multirank_gpu.py tests/test_bench_shapes_gpu.py -q -x > gpurun_out/r5u_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5u_pytest.log )
tail -5 gpurun_out/r5u_pytest.log
