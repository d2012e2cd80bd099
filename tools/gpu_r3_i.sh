#!/bin/bash
# round 3, session I: full GPU suite, the bench line (all legs), the temporal-reasoning edit, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/i_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/i_pytest.log
grep -v "amdgpu.ids\|Gloo\|socket.cpp" gpurun_out/i_pytest.log | grep "^E \|passed\|failed\|FAILED" | head -20
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/i_bench.json
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp8-leg --reasoning-edit --reasoning-steps 10 2>/dev/null | tail -1 > gpurun_out/i_bench_reasoning.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/i_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-edit --no-fp8-leg > $GRAFT_REPO_ROOT/gpurun_out/i_rocprof.log 2>&1)
python - <<'PY'
import json
for f in ("gpurun_out/i_bench.json", "gpurun_out/i_bench_reasoning.json"):
    try:
        d = json.loads(open(f).read())
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "host_enqueue_ms_per_step", "steps_per_sec_with_context_kv_cache", "steps_per_sec_fp8_mode", "vae", "sec_per_edit", "sec_per_edit_temporal_reasoning", "roofline_family")})
        print("  roofline", d.get("roofline"))
        print("  cpu", d.get("cpu_baseline"), d.get("cpu_config0"))
    except Exception as e:
        print(f, "unreadable", e)
PY
