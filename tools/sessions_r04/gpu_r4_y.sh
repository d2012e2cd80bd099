#!/bin/bash
# round-4 validation, part 3: smoke() as the driver calls it + the default bench line on the final tree
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
SECONDS=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
echo "smoke after ${SECONDS}s"
timeout 1500 python bench.py > gpurun_out/r4y_bench.json 2> gpurun_out/r4y_bench.err
echo "bench exit $? after ${SECONDS}s"
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r4y_bench.json").read().strip().splitlines()[-1])
print({k: o[k] for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "steps_per_sec_with_context_kv_cache", "steps_per_sec_fp8_mode")})
print("roofline", {k: o["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")})
print("family", o["roofline_family"]["achieved"], o["roofline_family"]["frac"])
print("sec_per_edit", {k: v["seconds"] if isinstance(v, dict) else v for k, v in o["sec_per_edit"].items()})
print("reasoning", {k: v["seconds"] for k, v in (o.get("sec_per_edit_temporal_reasoning") or {}).items()})
print("vae", o.get("vae"))
PY
