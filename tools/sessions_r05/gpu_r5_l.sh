#!/bin/bash
# round 5, session l: the fp8 STEP with the fp8 GEMM's round-4 LDS-DMA schedule (F8_DMA_SCHED=0 build) against the round-5 one, same box, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--fp8 --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit"
: > gpurun_out/r5l_fp8_step_ab.txt
for rep in 1 2; do
  for v in 0 1; do
    r=$(CE_HIPLIB_PATH=$PWD/chronoedit_amd/lib/libce_sched$v.so timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep  F8_DMA_SCHED=$v  720p fp8 step: steps/s, ms/step = $r" | tee -a gpurun_out/r5l_fp8_step_ab.txt
  done
done
for v in 0 1; do
  r=$(CE_HIPLIB_PATH=$PWD/chronoedit_amd/lib/libce_sched$v.so timeout 300 python bench.py $F --height 1056 --width 1584 --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "F8_DMA_SCHED=$v  1584x1056 fp8 step (configs[4]): steps/s, ms/step = $r" | tee -a gpurun_out/r5l_fp8_step_ab.txt
done
