#!/bin/bash
# The other BASELINE.json shapes on the current kernels (one bench line each, secondary legs off): gpurun_out/configs.log
F="--no-vae --no-encoders --no-fp8-leg --no-cpu-baseline --no-profile"
mkdir -p gpurun_out; : > gpurun_out/configs.log
run() { echo "== $*" >> gpurun_out/configs.log; timeout 600 python bench.py $F "$@" 2>/dev/null | tail -1 | cut -c1-900 >> gpurun_out/configs.log; }
run --guidance 1.0 --steps 8 --warmup 1                      # configs[2]: distilled, one forward per step
run --guidance 1.0 --steps 8 --warmup 1 --graph              # same, hipGraph replay
run --frames 8 --steps 2 --warmup 1                          # configs[3] shape on one GPU (N = 28 800)
run --height 1056 --width 1584 --steps 3 --warmup 1          # configs[4] resolution, bf16
run --height 1056 --width 1584 --steps 3 --warmup 1 --fp8    # configs[4]: fp8 GEMM mode
run --fp8 --steps 4 --warmup 1                               # 720p, fp8 GEMM mode
python - <<'PY'
import json
for l in open("gpurun_out/configs.log"):
    if l.startswith("=="):
        print(l.strip()); continue
    try:
        i = l.index('"achieved_tflops_per_gpu"'); print("   value", json.loads(l[:l.index(', "unit"')] + "}")["value"], l[i:i + 45])
    except Exception as e:
        print("   ?", l[:200])
PY
