#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mxfp8_gemm_gpu.py tests/test_fp8_gpu.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4n_pytest_mx.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4n_pytest_mx.log
grep -v amdgpu.ids gpurun_out/r4n_pytest_mx.log | tail -8
timeout 600 python bench.py --fp8 --steps 6 --warmup 2 --no-vae --no-encoders --no-edit --no-cpu-baseline > gpurun_out/r4n_bench_fp8_mx.json 2> gpurun_out/r4n_bench_fp8_mx.err
timeout 600 python bench.py --fp8 --fp8-row-scales --steps 6 --warmup 2 --no-vae --no-encoders --no-edit --no-cpu-baseline > gpurun_out/r4n_bench_fp8_row.json 2> gpurun_out/r4n_bench_fp8_row.err
python - <<'PY'
import json
for f in ("gpurun_out/r4n_bench_fp8_mx.json", "gpurun_out/r4n_bench_fp8_row.json"):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, o["value"], o["ms_per_step"], o["dtype"][:60])
        for k, v in list(o["kernel_breakdown"].items())[:12]:
            print("   ", k, v["n"], v["avg_ms"], v["tflops"], v["GBps"])
    except Exception as e:
        print(f, "failed", e)
PY
