#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gemm_variants.py 7,6,4 5 2>/dev/null | grep -v amdgpu.ids > gpurun_out/l_gemm_variants.log; cat gpurun_out/l_gemm_variants.log | cut -c1-400
timeout 500 python -m pytest tests/test_hip_kernels.py -k "gemm" -q --no-header -p no:cacheprovider > gpurun_out/l_pytest.log 2>&1; grep -v amdgpu.ids gpurun_out/l_pytest.log | tail -3
for i in 1 2; do
for v in -2 auto; do
  if [ $v = auto ]; then unset CE_GEMM_VARIANT; else export CE_GEMM_VARIANT=$v; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/l_bench_${v}_$i.log
  python - <<PY
import json
d = json.loads(open("gpurun_out/l_bench_${v}_$i.log").read())
kb = d["kernel_breakdown"]
print("variant $v run $i", d["value"], d["ms_per_step"], d["roofline_family"]["total_ms"], {k: round(kb[k]["avg_ms"], 4) for k in ("gemm_14400x5120x5120_epi2", "gemm_14400x13824x5120_epi1", "gemm_14400x5120x13824_epi2", "gemm_14400x10240x5120_epi0", "gemm_14400x5120x5120_epi0", "attention_7200x7200+0_h40_b2")})
PY
done
done
