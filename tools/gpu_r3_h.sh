#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/h_rccl_probe.log
for args in "5 thread_local" "5 global" "6 thread_local"; do
  echo "=== probe args: $args" >> gpurun_out/h_rccl_probe.log
  MASTER_PORT=$((29800+RANDOM%100)) timeout 150 python tools/rccl_graph_probe.py $args 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^frame #" | head -30 >> gpurun_out/h_rccl_probe.log
  echo "exit ${PIPESTATUS[0]}" >> gpurun_out/h_rccl_probe.log
done
cat gpurun_out/h_rccl_probe.log
timeout 900 python -m pytest tests/test_ulysses.py -m gpu -q --no-header -p no:cacheprovider -k "rccl or sharing_one_gpu or sharded_over_ranks" > gpurun_out/h_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/h_pytest.log
grep -v "amdgpu.ids\|Gloo\|socket.cpp" gpurun_out/h_pytest.log | grep "^E \|passed\|failed\|FAILED\|Error" | head -20
