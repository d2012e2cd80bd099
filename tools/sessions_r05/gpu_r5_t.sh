#!/bin/bash
# round 5, session t: the tests the last changes touch (lazy graph capture, fp8 GEMM source) + the default bench line of the final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_pipeline_gpu.py tests/test_mxfp8_gemm_gpu.py tests/test_fp8_gpu.py tests/test_ref_loop_gpu.py tests/test_run_inference_main_gpu.py tests/test_ulysses.py -q -x -k "not world_8 and not ranks_sharing" > gpurun_out/r5t_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5t_pytest.log )
tail -6 gpurun_out/r5t_pytest.log
timeout 900 python bench.py > gpurun_out/r5t_bench_default.json 2> gpurun_out/r5t_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5t_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value","ms_per_step","steps_per_sec_fp8_mode","fp8_mode_frac_of_fp8_peak","mfma_roofline_frac_whole_step")})
print(d["steps_per_sec_fp8_config4"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["roofline"]["frac"], d["roofline"]["traffic_source"])
print(d["sec_per_edit"]); print(d["sec_per_edit_temporal_reasoning"])
PY
