#!/bin/bash
# round 5, session x: energy per flop of the step's kernels (power-limited step: what decides its time) + the bare-MFMA floors under the same reading
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5x_kernel_power.txt
timeout 300 python tools/kernel_power.py > $O 2>&1
for p in mfma_rate_probe mfma_rate_probe_fp8; do
  [ -x tools/probes/$p ] || continue
  ( for i in 1 2 3 4 5 6; do tools/probes/$p > /tmp/$p.out 2>&1; done ) &
  BP=$!
  sleep 1.0; a=$(rocm-smi --showpower --showclocks | grep -E "Package Power|sclk" | tr '\n' ' ' | sed 's/  */ /g'); sleep 0.7; b=$(rocm-smi --showpower --showclocks | grep -E "Package Power|sclk" | tr '\n' ' ' | sed 's/  */ /g')
  wait $BP
  echo "== tools/probes/$p running (mixed zero / random operand phases; two readings):" >> $O; echo "   $a" >> $O; echo "   $b" >> $O; tail -4 /tmp/$p.out >> $O
done
cat $O | cut -c1-230
