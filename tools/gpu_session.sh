#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.log").read())
print({k: d[k] for k in ("value", "ms_per_step", "steps", "achieved_tflops_per_gpu", "vae", "steps_per_sec_with_context_kv_cache", "steps_per_sec_fp8_mode") if k in d})
print(d["sec_per_edit"], d["roofline"]["frac"], d["roofline_family"]["frac"])
PY
