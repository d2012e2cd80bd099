#!/bin/bash
# round 5, session e: the whole GPU suite (with durations) + smoke on the tree with the new fp8 GEMM schedule and the new bench legs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu --durations=30 -x > gpurun_out/r5e_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5e_pytest.log )
tail -45 gpurun_out/r5e_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
