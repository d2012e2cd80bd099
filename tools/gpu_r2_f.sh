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
timeout 300 python tools/microbench.py attn attn8 row 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench_r2.txt
timeout 600 python bench.py --fp8 --height 1056 --width 1584 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_1584.json
timeout 600 python bench.py --fp8 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_fp8_720p.json
timeout 300 python tools/full_edit.py --steps 50 2>/dev/null | tail -1 > gpurun_out/full_edit_50.json
timeout 200 python tools/full_edit.py --steps 8 --guidance 1.0 2>/dev/null | tail -1 > gpurun_out/full_edit_8.json
timeout 400 python bench.py --height 1056 --width 1584 --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_bf16_1584.json
timeout 600 python bench.py --frames 8 --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-encoders --no-edit --no-fp8-leg 2>/dev/null | tail -1 > gpurun_out/bench_n28800.json
cat gpurun_out/pmc_attn_vt_7200_b2.txt gpurun_out/pmc_attn_mxfp8_7200_b2.txt | grep -v "^ *SQ_" | head -60
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "sec_per_edit", "roofline_family", "steps_per_sec_fp8_mode")})
print(d["roofline"])
print(d["cpu_baseline"])
for f in ("bench_fp8_1584", "bench_fp8_720p", "bench_bf16_1584", "bench_n28800", "full_edit_50", "full_edit_8"):
    try:
        e = json.load(open(f"gpurun_out/{f}.json"))
        print(f, {k: e.get(k) for k in ("value", "ms_per_step", "seconds", "sec_per_edit") if k in e}, (e.get("roofline") or {}).get("kernel"), (e.get("roofline") or {}).get("achieved"))
    except Exception as ex:
        print(f, "unreadable", ex)
PY
cat gpurun_out/microbench_r2.txt
