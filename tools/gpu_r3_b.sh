#!/bin/bash
# round 3, session B: whole step under the GEMM main loops, alternating runs on one box (noise check)
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
for v in 1 4 3; do
  CE_GEMM_VARIANT=$v timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/b_bench_v${v}_$i.log
  python - <<PY
import json
d = json.loads(open("gpurun_out/b_bench_v${v}_$i.log").read())
kb = d["kernel_breakdown"]
print("variant $v run $i", d["value"], d["ms_per_step"], d["roofline_family"]["total_ms"], {k: round(kb[k]["avg_ms"], 4) for k in ("ln_affine_14400x5120", "rmsnorm_rope_14400x5120", "gemm_14400x5120x5120_epi2", "gemm_14400x13824x5120_epi1", "attention_7200x7200+0_h40_b2")})
PY
done
done
