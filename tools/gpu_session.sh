#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/rank_emulation.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/s_rank_emulation.log; cat gpurun_out/s_rank_emulation.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-profile --reasoning-edit 2>/dev/null | tail -1 > gpurun_out/s_bench_reasoning.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s_bench_reasoning.log").read())
print({k: d.get(k) for k in ("value", "vae", "sec_per_edit", "sec_per_edit_temporal_reasoning")})
PY
