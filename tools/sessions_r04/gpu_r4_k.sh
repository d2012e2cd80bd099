#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
PROBE_MODE=eager8 timeout 120 python tools/owned_comm_probe.py > gpurun_out/r4k_owned_comm_probe_eager8.txt 2>&1
echo "exit $?" >> gpurun_out/r4k_owned_comm_probe_eager8.txt
grep -v amdgpu.ids gpurun_out/r4k_owned_comm_probe_eager8.txt | tail -8
timeout 200 python -m pytest tests/test_ulysses.py -m gpu -q --no-header -p no:cacheprovider -k "owned" -x > gpurun_out/r4k_pytest_comm.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4k_pytest_comm.log
grep -v amdgpu.ids gpurun_out/r4k_pytest_comm.log | tail -12
