#!/bin/bash
# round 6, session g: the register-direct epilogue ported to the bf16 large-tile GEMMs (ce_gemm384.hip, ce_gemm256w4.hip): parity, then the A/B
# against the round-5 kernels (libce_gemm_r5.so = this tree with the two files of HEAD~), stand-alone and inside the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_dit_forward_gpu.py tests/test_encoders_gpu.py tests/test_vae_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -5
L=chronoedit_amd/lib
timeout 900 python tools/gemm_ab.py $L/libce_gemm_r5.so $L/libchronoedit_hip.so 2>&1 | grep -v amdgpu.ids | tee $O/r6g_gemm_bf16_epilogue_ab.txt
F="--steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit --no-full-edit --no-reasoning-edit"
: > $O/r6g_bf16_step_epilogue_ab.txt
for rep in 1 2 3; do
  for v in ce_gemm_r5 chronoedit_hip; do
    r=$(CE_HIPLIB_PATH=$PWD/$L/lib$v.so timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep  $v  720p bf16 step (configs[1]): steps/s, ms/step = $r" | tee -a $O/r6g_bf16_step_epilogue_ab.txt
  done
done
for v in ce_gemm_r5 chronoedit_hip; do
  r=$(CE_HIPLIB_PATH=$PWD/$L/lib$v.so timeout 300 python bench.py $F --guidance 1.0 --steps 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$v  720p distilled step (configs[2], B = 1): steps/s, ms/step = $r" | tee -a $O/r6g_bf16_step_epilogue_ab.txt
done
