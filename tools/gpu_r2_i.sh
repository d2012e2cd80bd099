#!/bin/bash
# round-2 session I: V^T-from-the-projection path: kernel tests, model-level parity, bench A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_dit_forward_gpu.py tests/test_pipeline_gpu.py tests/test_ref_loop_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/pytest_i.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_i.log
tail -6 gpurun_out/pytest_i.log
for v in "" "--no-transposed-v" "" "--no-transposed-v"; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-edit $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
kb=d['kernel_breakdown']
print('$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_family']['achieved'], {k:(v['avg_ms'],v['tflops']) for k,v in list(kb.items())[:7]})
"
done
