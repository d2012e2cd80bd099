#!/bin/bash
# round-4 validation, part 2: the default bench line (as the driver runs it), its rocprofv3 kernel stats, the other configs, the VAE
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SECONDS=0
timeout 1500 python bench.py > gpurun_out/r4r_bench.json 2> gpurun_out/r4r_bench.err
echo "bench exit $? after ${SECONDS}s"
python - <<'PY'
import json
o = json.loads(open("gpurun_out/r4r_bench.json").read().strip().splitlines()[-1])
print({k: o[k] for k in ("value", "ms_per_step", "achieved_tflops_per_gpu", "steps_per_sec_with_context_kv_cache", "steps_per_sec_fp8_mode")})
print("roofline", {k: o["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "traffic_source")})
print("family", o["roofline_family"]["achieved"], o["roofline_family"]["frac"], o["roofline_family"]["share_of_step"])
print("sec_per_edit", json.dumps(o["sec_per_edit"])[:1200])
print("reasoning", json.dumps(o.get("sec_per_edit_temporal_reasoning"))[:1200])
print("vae", json.dumps(o.get("vae"))[:400]); print("cpu", json.dumps(o.get("cpu_baseline"))[:500])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4r_prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit --no-full-edit --no-reasoning-edit > $R/gpurun_out/r4r_rocprof.log 2>&1)
ls gpurun_out/r4r_prof/* | head; find gpurun_out/r4r_prof -name "*kernel_stats.csv" -exec head -25 {} \;
bash tools/gpu_configs.sh 2>&1 | tail -14
CE_VAE_BENCH_OUT=gpurun_out/r4r_vae_bench.json timeout 300 python tools/vae_bench.py 720 1280 5 > gpurun_out/r4r_vae_bench.log 2>&1
grep -E "^encode|^decode" gpurun_out/r4r_vae_bench.log
echo "total ${SECONDS}s"
