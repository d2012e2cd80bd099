#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/g_rccl_probe.log
for mode in thread_local relaxed; do
for st in 3 5; do
  echo "=== mode $mode, probe up to stage $st" >> gpurun_out/g_rccl_probe.log
  MASTER_PORT=$((29700+st)) timeout 150 python tools/rccl_graph_probe.py $st $mode 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^frame #" | head -40 >> gpurun_out/g_rccl_probe.log
  echo "exit ${PIPESTATUS[0]}" >> gpurun_out/g_rccl_probe.log
done
done
cat gpurun_out/g_rccl_probe.log
timeout 1500 python -m pytest tests/test_ulysses.py tests/test_run_inference_main_gpu.py tests/test_vae_gpu.py -m gpu -q --no-header -p no:cacheprovider -s > gpurun_out/g_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/g_pytest.log
grep -v "amdgpu.ids\|Gloo\|socket.cpp" gpurun_out/g_pytest.log | grep "^E \|passed\|failed\|FAILED\|Error\|hip \|world 8" | head -40
timeout 300 python tools/vae_bench.py > gpurun_out/g_vae_bench.log 2>&1; grep -v amdgpu.ids gpurun_out/g_vae_bench.log | head -8
