#!/bin/bash
# round 5, session o: per-kernel times INSIDE the fp8 step for the two LDS-DMA schedules of the fp8 GEMM (bench.py's HIP-event profile of one step), same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-edit"
: > gpurun_out/r5o_fp8_step_kernels_ab.txt
for rep in 1 2; do
for v in 0 1; do
  CE_HIPLIB_PATH=$PWD/chronoedit_amd/lib/libce_sched$v.so timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
kb=d['kernel_breakdown']
print('rep $rep F8_DMA_SCHED=$v: %.4f steps/s, %.2f ms/step' % (d['value'], d['ms_per_step']))
for k,v in kb.items():
    if k.startswith(('gemm_mxfp8','attention_mxfp8')): print('    %-46s n=%3d avg %.4f ms  %s TF' % (k, v['n'], v['avg_ms'], v['tflops']))
" | tee -a gpurun_out/r5o_fp8_step_kernels_ab.txt
done
done
