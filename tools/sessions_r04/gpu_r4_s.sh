#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
F="--no-vae --no-encoders --no-edit --no-full-edit --no-reasoning-edit --no-cpu-baseline --height 1056 --width 1584 --steps 3 --warmup 1"
timeout 600 python bench.py --fp8 $F > gpurun_out/r4s_fp8_1584_mx.json 2> gpurun_out/r4s_mx.err
timeout 600 python bench.py --fp8 --fp8-row-scales $F > gpurun_out/r4s_fp8_1584_row.json 2> gpurun_out/r4s_row.err
python - <<'PY'
import json
for f in ("gpurun_out/r4s_fp8_1584_mx.json", "gpurun_out/r4s_fp8_1584_row.json"):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, o["value"], o["ms_per_step"], o["dtype"][:60])
        for k, v in list(o["kernel_breakdown"].items())[:14]:
            print("   ", k, v["n"], v["avg_ms"], v["tflops"], v["GBps"])
    except Exception as e:
        print(f, "failed", e)
PY
