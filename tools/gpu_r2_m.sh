#!/bin/bash
# round-2 session M: persistent V^T attention - attention tests, model-level tests, bench A/B vs the previous commit is in profiles
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_dit_forward_gpu.py tests/test_ulysses.py tests/test_encoders_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "attention or forward or ulysses or denoise or clip or umt5" > gpurun_out/pytest_m.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_m.log
tail -4 gpurun_out/pytest_m.log
for i in 1 2; do
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-edit 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
kb=d['kernel_breakdown']
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_family']['achieved'], {k:(v['avg_ms'],v['tflops']) for k,v in list(kb.items())[:3]})
"
done
timeout 300 python tools/microbench.py attn 2>&1 | grep -v amdgpu.ids | grep "V^T"
