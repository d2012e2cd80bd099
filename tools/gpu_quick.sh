#!/bin/bash
# quick GPU iteration: selected tests + selected microbench.  usage: gpu_quick.sh "<pytest -k expr>" "<microbench args>"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -k "$1" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
[ -n "$2" ] && timeout 600 python tools/microbench.py $2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/microbench.log
