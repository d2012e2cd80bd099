#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -v amdgpu.ids gpurun_out/pytest_gpu.log | tail -6
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench.log
timeout 600 python bench.py --steps 6 --warmup 2 --graph --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_graph.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-edit --no-fp8-leg > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1)
python - <<'PY'
import json
for f in ("gpurun_out/bench.log", "gpurun_out/bench_graph.log"):
    try:
        d = json.loads(open(f).read())
        print(f, {k: d[k] for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "launch", "vae", "sec_per_edit_8_steps_measured", "sec_per_edit_50_steps", "roofline", "roofline_family", "cpu_baseline") if k in d})
    except Exception as e:
        print(f, "unreadable", e)
PY
bash tools/gpu_pmc_traffic.sh gemm384_14400x5120x13824 gemm 14400 5120 13824 2 6 5 > /dev/null 2>&1
cat gpurun_out/pmc_gemm384_14400x5120x13824.txt | head -40
