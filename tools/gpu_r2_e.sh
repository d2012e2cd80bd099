#!/bin/bash
# round-2 session E: MXFP8 attention parity + producer timings + fp8-mode bench lines
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_ulysses.py -m gpu -q --no-header -p no:cacheprovider -x -s > gpurun_out/pytest_e.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_e.log
grep -E "passed|failed|mxfp8|DiT|Error|assert|rccl|RCCL" gpurun_out/pytest_e.log | tail -14
timeout 300 python tools/microbench.py attn8 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench_attn8.txt; cat gpurun_out/microbench_attn8.txt
timeout 600 python bench.py --fp8 --height 1056 --width 1584 --steps 4 --warmup 1 2>/dev/null | tail -1 > gpurun_out/bench_fp8_1584.json
timeout 600 python bench.py --fp8 --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_fp8_720p.json
python - <<'PY'
import json
for f in ("bench_fp8_1584", "bench_fp8_720p"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, {k: d.get(k) for k in ("value", "ms_per_step", "dtype", "sec_per_edit")})
    print(d["roofline"])
    for k, v in list(d["kernel_breakdown"].items())[:10]:
        print("  ", k, v)
PY
