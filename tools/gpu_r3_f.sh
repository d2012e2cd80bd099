#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in 2 3 4 5; do
  echo "=== probe up to stage $st" >> gpurun_out/f_rccl_probe.log
  MASTER_PORT=$((29600+st)) timeout 150 python tools/rccl_graph_probe.py $st >> gpurun_out/f_rccl_probe.log 2>&1
  echo "exit $?" >> gpurun_out/f_rccl_probe.log
done
grep -v "amdgpu.ids\|socket.cpp" gpurun_out/f_rccl_probe.log | tail -60
timeout 1200 python -m pytest tests/test_run_inference_main_gpu.py tests/test_ref_loop_gpu.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/f_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/f_pytest.log
grep -v "amdgpu.ids\|Gloo\|socket.cpp" gpurun_out/f_pytest.log | grep "^E \|passed\|failed\|FAILED\|Error" | head -40
timeout 300 python tools/vae_bench.py > gpurun_out/f_vae_bench.log 2>&1; tail -40 gpurun_out/f_vae_bench.log
