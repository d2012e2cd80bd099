#!/bin/bash
# round-4 validation, part 1: the whole GPU suite as the driver runs it
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
SECONDS=0
timeout 2400 python -m pytest tests -m gpu -x -q --no-header -p no:cacheprovider --durations=15 > gpurun_out/r4q_pytest_gpu.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/r4q_pytest_gpu.log
grep -v amdgpu.ids gpurun_out/r4q_pytest_gpu.log | tail -30
