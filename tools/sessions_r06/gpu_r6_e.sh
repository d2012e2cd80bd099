#!/bin/bash
# round 6, session e: (1) fp8 accuracy at the full width, Linear by Linear (tools/fp8_sensitivity.py); (2) the fp8 STEP with the staged epilogue
# build against the register-direct one, alternating; (3) fp8 + DiT tests on the new engine switches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python tools/fp8_sensitivity.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|amax = " | tee $O/r6e_fp8_sensitivity.txt
F="--fp8 --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit --no-full-edit --no-reasoning-edit"
: > $O/r6e_fp8_step_epilogue_ab.txt
for rep in 1 2 3; do
  for v in ce_f8epilds chronoedit_hip; do
    r=$(CE_HIPLIB_PATH=$PWD/chronoedit_amd/lib/lib$v.so timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep  $v  720p fp8 step: steps/s, ms/step = $r" | tee -a $O/r6e_fp8_step_epilogue_ab.txt
  done
done
for v in ce_f8epilds chronoedit_hip; do
  r=$(CE_HIPLIB_PATH=$PWD/chronoedit_amd/lib/lib$v.so timeout 300 python bench.py $F --height 1056 --width 1584 --steps 6 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$v  1584x1056 fp8 step (configs[4]): steps/s, ms/step = $r" | tee -a $O/r6e_fp8_step_epilogue_ab.txt
done
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_dit_forward_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -5
