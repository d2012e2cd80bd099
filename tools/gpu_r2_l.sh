#!/bin/bash
# round-2 session L: the other BASELINE shapes on the final build + the driver's own invocation, timed
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_driver_style.json; t1=$(date +%s); echo "driver-style bench wall: $((t1-t0)) s"
timeout 300 python bench.py --guidance 1 --steps 8 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_distilled.json
timeout 300 python bench.py --graph --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_graph.json
timeout 300 python bench.py --cache-context --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_ctxcache.json
python - <<'PY'
import json
for f in ("bench_driver_style", "bench_distilled", "bench_graph", "bench_ctxcache"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["config"]["workload"][:90], d.get("launch"), d["roofline"]["frac"] if d.get("roofline") else None)
    except Exception as e:
        print(f, "unreadable", e)
PY
