#!/bin/bash
# round 6, session p: the full-width oracle evaluations on the device (oracle/device.py) vs on the host cores: same figures? how much wall time?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
K="full_width_block or full_width_eight or configs0_edit"
for dev in cuda cpu; do
  echo "== CE_ORACLE_DEVICE=$dev" | tee -a $O/r6p_oracle_device.txt
  ( time CE_ORACLE_DEVICE=$dev timeout 1200 python -m pytest tests/test_width_depth_gpu.py tests/test_bench_shapes_gpu.py tests/test_dit_forward_gpu.py -q -m gpu -s -k "$K" --durations=8 2>&1 \
      | grep "rel-L2\|vs fp32\|fp8 mode\|passed\|failed\|Error\|error\|s call" ) 2>&1 | tee -a $O/r6p_oracle_device.txt
done
