#!/bin/bash
# round-2 session O: the bench lines of the final build (default, fp8 at both resolutions)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench.json
timeout 400 python bench.py --fp8 --height 1056 --width 1584 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_1584.json
timeout 400 python bench.py --fp8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_720p.json
python - <<'PY'
import json
for f in ("bench", "bench_fp8_1584", "bench_fp8_720p"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], (d.get("roofline_family") or {}).get("frac"), d.get("steps_per_sec_fp8_mode"))
PY
