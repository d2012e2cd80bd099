#!/bin/bash
# round 6, session y: the 288-row tile in the step: GEMM / forward parity suites, then the distilled B = 1 step (configs[2]) and the headline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_dit_forward_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/r6y_pytest.txt
F="--no-vae --no-encoders --no-fp8-leg --no-fp8-config4 --no-cpu-baseline --no-edit --no-full-edit --no-reasoning-edit"
: > $O/r6y_configs.jsonl
run() { echo "# bench.py $*" >> $O/r6y_configs.jsonl; timeout 600 python bench.py $F "$@" 2>>$O/r6y_err.log | tail -1 >> $O/r6y_configs.jsonl; echo "rc $? $*"; }
run --guidance 1.0 --steps 16 --warmup 2
run --guidance 1.0 --steps 16 --warmup 2 --graph
run --steps 10 --warmup 2
run --guidance 1.0 --steps 16 --warmup 2
python - <<'PY'
import json
for l in open("gpurun_out/r6y_configs.jsonl"):
    if l.startswith("#"):
        print(l.strip()); continue
    d = json.loads(l)
    print("  ", d["value"], d["ms_per_step"], d.get("mfma_roofline_frac_whole_step"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
    kb = d.get("kernel_breakdown", {})
    print("     ", {k: (v["avg_ms"], v["tflops"]) for k, v in kb.items() if k.startswith("gemm_7200x5120x5120") or k.startswith("gemm_14400x5120x5120")})
PY
