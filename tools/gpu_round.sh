#!/bin/bash
# One GPU-box session: parity tests, microbench, bench line, rocprof kernel stats. Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/env.log 2>&1
import torch, os
print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None, os.cpu_count())
PY
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
if [ "$1" != "tests" ]; then
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1)
ls -R gpurun_out/prof | head -30 >> gpurun_out/rocprof.log
fi
tail -5 gpurun_out/pytest_gpu.log; tail -12 gpurun_out/microbench.log; tail -3 gpurun_out/bench.log
