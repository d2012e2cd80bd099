#!/bin/bash
# round 3, session A: the one-wave-per-SIMD GEMM main loop - parity, then A/B against the 8-wave loop, then the whole step under it
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py -k "gemm" -q --no-header -p no:cacheprovider -x > gpurun_out/a_pytest_gemm.log 2>&1
echo "pytest exit $?" >> gpurun_out/a_pytest_gemm.log
tail -5 gpurun_out/a_pytest_gemm.log
timeout 600 python tools/gemm_variants.py 1,3,4 5 2>&1 | grep -v amdgpu.ids > gpurun_out/a_gemm_variants.log
cat gpurun_out/a_gemm_variants.log
for v in 1 3; do
  CE_GEMM_VARIANT=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/a_bench_v$v.log
  python - <<PY
import json
d = json.loads(open("gpurun_out/a_bench_v$v.log").read())
print("variant $v", {k: d.get(k) for k in ("value", "ms_per_step", "roofline_family")})
PY
done
