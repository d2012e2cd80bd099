#!/bin/bash
# session J: the sharded step on the library-owned RCCL communicator, captured (one rank), and the other one-rank RCCL tests
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_ulysses.py -m gpu -q --no-header -p no:cacheprovider -k "owned or eagerly or one_rank" -x > gpurun_out/r4j_pytest_comm.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4j_pytest_comm.log
grep -v amdgpu.ids gpurun_out/r4j_pytest_comm.log | tail -40
