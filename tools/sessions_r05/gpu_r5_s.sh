#!/bin/bash
# round 5, session s: per-kernel times inside the fp8 step, two-stage loop vs A ring of three (F8_A3), same box, alternating, three repetitions
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--fp8 --steps 8 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-edit"
: > gpurun_out/r5s_fp8_step_a3_kernels.txt
for rep in 1 2 3; do
for v in 0 1; do
  CE_HIPLIB_PATH=$PWD/chronoedit_amd/lib/libce_a3_$v.so timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
kb=d['kernel_breakdown']
g=sum(v['avg_ms']*v['n'] for k,v in kb.items() if k.startswith('gemm_mxfp8'))
a=sum(v['avg_ms']*v['n'] for k,v in kb.items() if k.startswith('attention'))
o=sum(v['avg_ms']*v['n'] for k,v in kb.items() if not k.startswith(('attention','gemm_mxfp8')))
print('rep $rep F8_A3=$v: %.4f steps/s, %.2f ms/step | profiled step: fp8 GEMMs %.2f ms, attention %.2f ms, everything else %.2f ms' % (d['value'], d['ms_per_step'], g, a, o))
for k,v in kb.items():
    if k.startswith('gemm_mxfp8'): print('    %-46s n=%3d avg %.4f ms  %s TF' % (k, v['n'], v['avg_ms'], v['tflops']))
" | tee -a gpurun_out/r5s_fp8_step_a3_kernels.txt
done
done
