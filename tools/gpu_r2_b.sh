#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gemm_slab_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_slab_ab.txt
cat gpurun_out/gemm_slab_ab.txt
timeout 600 python bench.py --frames 8 --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_n28800.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_n28800.json"))
print({k: d[k] for k in ("value", "ms_per_step", "achieved_tflops_per_gpu")}, d["config"]["workload"])
for k, v in list(d["kernel_breakdown"].items())[:8]:
    print(k, v)
PY
