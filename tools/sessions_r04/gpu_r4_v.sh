#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
F="--no-vae --no-encoders --no-edit --no-full-edit --no-reasoning-edit --no-cpu-baseline --steps 6 --warmup 2"
for i in 1 2; do
timeout 600 python bench.py --fp8 $F > gpurun_out/r4v_fp8_fused_$i.json 2> gpurun_out/r4v_fused.err
timeout 600 python bench.py --fp8 --fp8-no-attn-quant-fusion $F > gpurun_out/r4v_fp8_unfused_$i.json 2> gpurun_out/r4v_unfused.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4v_fp8_*.json")):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        kb = o["kernel_breakdown"]
        print(f, o["value"], o["ms_per_step"], [(k, v["avg_ms"]) for k, v in kb.items() if k.startswith("attention") or k.startswith("quant")])
    except Exception as e:
        print(f, "failed", e)
PY
