#!/bin/bash
# round-2 session A: parity suite, L2 pattern probe, bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|error|rel-L2|trajectory|reference __call__" gpurun_out/pytest_gpu.log | tail -30
timeout 120 tools/probes/l2_pattern_probe > gpurun_out/l2_pattern_probe.txt 2>&1
cat gpurun_out/l2_pattern_probe.txt
timeout 900 python bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "sec_per_edit", "roofline_family", "steps_per_sec_fp8_gemm_mode")})
print(d["roofline"])
print(d["cpu_baseline"])
for k, v in list(d["kernel_breakdown"].items())[:12]:
    print(k, v)
PY
