#!/bin/bash
# round-2 session K: fp8 GEMM with the W-first MFMA operand order (vector epilogue): parity, microbench, fp8-mode bench lines, smoke
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/pytest_k.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_k.log
tail -3 gpurun_out/pytest_k.log
timeout 300 python tools/microbench.py fp8 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench_fp8.txt; cat gpurun_out/microbench_fp8.txt
timeout 600 python bench.py --fp8 --height 1056 --width 1584 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_1584.json
timeout 600 python bench.py --fp8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_720p.json
python - <<'PY'
import json
for f in ("bench_fp8_1584", "bench_fp8_720p"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
    for k, v in list(d["kernel_breakdown"].items())[:8]:
        print("  ", k, v["avg_ms"], v["tflops"])
PY
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
