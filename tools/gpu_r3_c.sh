#!/bin/bash
# round 3, session C: W4 GEMM with the prefetched gated-residual epilogue + one-barrier variant; new bench-shape parity tests; PMC v1 vs v4
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_ulysses.py -k "gemm" -q --no-header -p no:cacheprovider -x > gpurun_out/c_pytest_gemm.log 2>&1
echo "pytest exit $?" >> gpurun_out/c_pytest_gemm.log
tail -5 gpurun_out/c_pytest_gemm.log
timeout 600 python tools/gemm_variants.py 1,4,3,5 5 2>&1 | grep -v amdgpu.ids > gpurun_out/c_gemm_variants.log
cat gpurun_out/c_gemm_variants.log
timeout 1200 python -m pytest tests/test_bench_shapes_gpu.py -q --no-header -p no:cacheprovider -s > gpurun_out/c_pytest_shapes.log 2>&1
echo "pytest exit $?" >> gpurun_out/c_pytest_shapes.log
grep -v amdgpu.ids gpurun_out/c_pytest_shapes.log | tail -25
bash tools/gpu_pmc_traffic.sh c_gemm_v1 gemm 14400 13824 5120 1 1 3 > /dev/null 2>&1
bash tools/gpu_pmc_traffic.sh c_gemm_v4 gemm 14400 13824 5120 1 4 3 > /dev/null 2>&1
cat gpurun_out/pmc_c_gemm_v1.txt gpurun_out/pmc_c_gemm_v4.txt
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/c_bench.log
python - <<PY
import json
d = json.loads(open("gpurun_out/c_bench.log").read())
print("bench", d["value"], d["ms_per_step"], d["roofline_family"])
PY
