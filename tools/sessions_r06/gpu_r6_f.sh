#!/bin/bash
# round 6, session f: mixed-precision tests, the whole-edit fp8 figures at the full width, peaked-logit attention rates, then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -m gpu -k "mixed_precision or fp8_mode_vs_fp8_contract" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_width_depth_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu -s -k "configs0_edit or bf16_and_fp8_modes" 2>&1 | grep -v "Warning\|amax = \|^$" | tail -12 | tee $O/r6f_fp8_full_width.txt
timeout 900 python tools/attn_peaked.py 2>&1 | grep -v amdgpu.ids | tee $O/r6f_attn_peaked.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6f_bench.json 2> $O/r6f_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6f_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps_per_sec_fp8_mode'], d['fp8_mode_frac_of_fp8_peak'], d['steps_per_sec_fp8_config4']['value'], d['roofline']['kernel'], d['roofline']['frac'])
print(json.dumps(d['fp8_policies'])[:600])
print({k: v['seconds'] for k, v in d['sec_per_edit'].items() if isinstance(v, dict)}, d['vae'])
PY
