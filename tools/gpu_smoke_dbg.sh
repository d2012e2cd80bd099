#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke only"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke1.log 2>&1; echo "rc $?"; tail -8 gpurun_out/smoke1.log
echo "== build + smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke2.log 2>&1; echo "rc $?"; tail -12 gpurun_out/smoke2.log
