#!/bin/bash
# round-4 GPU session A: the tests touched this round + one bench line (no long edits)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_width_depth_gpu.py tests/test_vae_gpu.py tests/test_dit_forward_gpu.py tests/test_pipeline_gpu.py tests/test_hip_kernels.py tests/test_bench_shapes_gpu.py tests/test_run_inference_main_gpu.py tests/test_ref_loop_gpu.py tests/test_adapters_gpu.py tests/test_fp8_gpu.py -m gpu -q --no-header -p no:cacheprovider -s -x > gpurun_out/r4a_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4a_pytest.log
grep -v "amdgpu.ids" gpurun_out/r4a_pytest.log | grep -E "rel-L2|hip |passed|failed|error|Error|exit|x\)|configs|D = 5120" | tail -40
timeout 1200 python -m pytest tests/test_bench_multirank_gpu.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r4a_pytest_multirank.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4a_pytest_multirank.log
tail -25 gpurun_out/r4a_pytest_multirank.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-reasoning-edit --no-full-edit > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
echo "bench exit $?"
python - <<'PY'
import json
try:
    o = json.loads(open("gpurun_out/r4a_bench.json").read().strip().splitlines()[-1])
    print({k: o[k] for k in ("value", "ms_per_step", "steps_per_sec_with_context_kv_cache", "steps_per_sec_fp8_mode", "vae", "sec_per_edit")})
    print(o["roofline"]["kernel"], o["roofline"]["frac"], o["roofline_family"]["frac"])
    for k, v in list(o["kernel_breakdown"].items())[:14]:
        print(k, v)
except Exception as e:
    print("bench parse failed", e)
    print(open("gpurun_out/r4a_bench.err").read()[-3000:])
PY
