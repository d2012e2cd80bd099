#!/bin/bash
# round 5, session v: package power and shader clock while the bf16 / fp8 step runs (rocm-smi sampled twice a second beside bench.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r5v_power_clock_trace.txt
: > $O
rocm-smi --showpower --showclocks --showtemp 2>&1 | head -40 >> $O
echo "=== idle sample above; max / caps:" >> $O
rocm-smi --showmaxpower --showclkfrq 2>&1 | grep -iv "^$" | head -40 >> $O
F="--steps 40 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-edit --no-profile"
for mode in "" "--fp8"; do
  echo "=== bench.py $mode $F : samples every 0.5 s" >> $O
  ( timeout 300 python bench.py $mode $F > gpurun_out/r5v_bench$mode.json 2>/dev/null ) &
  BP=$!
  for i in $(seq 1 70); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk|mclk|fclk" | tr '\n' ' ' | sed 's/  */ /g' >> $O; echo >> $O
    sleep 0.5
    kill -0 $BP 2>/dev/null || break
  done
  wait $BP
  tail -1 gpurun_out/r5v_bench$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ->', d['value'], 'steps/s', d['ms_per_step'], 'ms/step')" >> $O
done
tail -60 $O | cut -c1-260
