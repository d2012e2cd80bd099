#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_ref_loop_gpu.py tests/test_ulysses.py tests/test_bench_multirank_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "graph or replay or pipeline or edit or captured or reasoning or call or main or sharded_line" > gpurun_out/r4aa_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r4aa_pytest.log
grep -v amdgpu.ids gpurun_out/r4aa_pytest.log | tail -6
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-profile --reasoning-steps 10 > gpurun_out/r4aa_bench.json 2> gpurun_out/r4aa_bench.err
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r4aa_bench.json").read().strip().splitlines()[-1])
print(o["value"], {k: (v["seconds"] if isinstance(v, dict) else v) for k, v in o["sec_per_edit"].items()}, {k: v["seconds"] for k, v in (o.get("sec_per_edit_temporal_reasoning") or {}).items()})
PY
timeout 300 python bench.py --graph --steps 4 --warmup 1 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-profile --no-edit --no-full-edit --no-reasoning-edit 2>/dev/null | tail -1 | cut -c1-200
