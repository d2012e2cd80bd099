#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5y_bench_final_tree.json 2> gpurun_out/r5y_bench.err; echo "rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r5y_bench_final_tree.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['power'], d['steps_per_sec_fp8_mode'], d['fp8_mode_frac_of_fp8_peak'], d['steps_per_sec_fp8_config4']['value'], d['cpu_baseline']['kind'], d['roofline']['frac'], d['roofline_family']['frac'])
print(d['sec_per_edit']['configs[2] 8-step distilled schedule, guidance 1 (measured end to end)']['seconds'], d['sec_per_edit']['configs[1] 50 steps, guidance 5 (measured end to end)']['seconds'], [v['seconds'] for v in d['sec_per_edit_temporal_reasoning'].values()], d['vae'])"
