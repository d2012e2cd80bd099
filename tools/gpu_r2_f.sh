#!/bin/bash
# round-2 session F: whole GPU suite, default bench line under rocprofv3 --kernel-trace --stats, PMC traffic of the two attention kernels
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 6 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg > $R/gpurun_out/rocprof.log 2>&1)
ls gpurun_out/prof/* | head
bash tools/gpu_pmc_traffic.sh attn_vt_7200_b2 attnvt 7200 40 2 3 > /dev/null 2>&1
bash tools/gpu_pmc_traffic.sh attn_mxfp8_7200_b2 attn8 7200 40 2 3 > /dev/null 2>&1
cat gpurun_out/pmc_attn_vt_7200_b2.txt gpurun_out/pmc_attn_mxfp8_7200_b2.txt | grep -v "^ *SQ_" | head -60
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "sec_per_edit", "roofline_family", "steps_per_sec_fp8_mode")})
print(d["roofline"])
print(d["cpu_baseline"])
PY
