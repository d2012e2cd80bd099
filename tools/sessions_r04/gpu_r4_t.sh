#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_mxfp8_gemm_gpu.py tests/test_fp8_gpu.py tests/test_hip_kernels.py -m gpu -q --no-header -p no:cacheprovider -x -k "quantised or fused or mx or fp8 or 2seg or cross" > gpurun_out/r4t_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4t_pytest.log
grep -v amdgpu.ids gpurun_out/r4t_pytest.log | tail -12
F="--no-vae --no-encoders --no-edit --no-full-edit --no-reasoning-edit --no-cpu-baseline --steps 6 --warmup 2"
timeout 600 python bench.py --fp8 $F > gpurun_out/r4t_fp8_fused.json 2> gpurun_out/r4t_fused.err
python - <<'PY'
import json
for f in ("gpurun_out/r4t_fp8_fused.json",):
    try:
        o = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, o["value"], o["ms_per_step"], o["dtype"][:60])
        for k, v in list(o["kernel_breakdown"].items())[:14]:
            print("   ", k, v["n"], v["avg_ms"], v["tflops"], v["GBps"])
    except Exception as e:
        print(f, "failed", e)
PY
