#!/bin/bash
# One GPU-box session: parity tests, microbench, bench lines (eager + graph), rocprof kernel stats. Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
if [ "$1" != "tests" ]; then
timeout 600 python tools/microbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench.log
timeout 900 python bench.py --steps 4 --warmup 1 2>/dev/null | tail -1 > gpurun_out/bench.log
timeout 900 python bench.py --steps 4 --warmup 1 --graph --no-cpu-baseline --no-vae --no-encoders 2>/dev/null | tail -1 > gpurun_out/bench_graph.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1)
ls -R gpurun_out/prof | head -20 >> gpurun_out/rocprof.log
cat gpurun_out/microbench.log
python - <<'PY'
import json
for f in ("gpurun_out/bench.log", "gpurun_out/bench_graph.log"):
    try:
        d = json.loads(open(f).read())
        print(f, {k: d[k] for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "launch", "vae", "sec_per_edit_50_steps", "roofline") if k in d})
    except Exception as e:
        print(f, "unreadable", e)
PY
fi
