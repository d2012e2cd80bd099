#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 6 --warmup 2 --no-reasoning-edit --no-full-edit --no-cpu-baseline > gpurun_out/r5w_bench.json 2> gpurun_out/r5w_bench.err; echo "rc $?"
python -c "
import json
d=json.loads(open('gpurun_out/r5w_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['power'], d['steps_per_sec_fp8_mode'], d['roofline']['frac'])"
