#!/bin/bash
# round 6, session j: the whole GPU suite (timed), smoke, the VAE account, the driver's bench command, complete lines for the other BASELINE configs
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=15 -x ) > $O/r6j_pytest.log 2>&1; tail -25 $O/r6j_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
CE_VAE_BENCH_OUT=$O/r6j_vae_bench_720p.json timeout 600 python tools/vae_bench.py 2>&1 | grep -v amdgpu.ids | tail -45
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6j_bench.json 2> $O/r6j_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6j_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['mfma_roofline_frac_whole_step'], d['steps_per_sec_fp8_mode'], d['fp8_mode_frac_of_fp8_peak'], d['steps_per_sec_fp8_config4']['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_family']['frac'])
print(json.dumps(d['fp8_policies'])[:400])
print({k: v['seconds'] for k, v in d['sec_per_edit'].items() if isinstance(v, dict)}, d['vae'], d['power'])
print(d['roofline']['by_symbol'])
PY
F="--no-vae --no-encoders --no-fp8-leg --no-fp8-config4 --no-cpu-baseline --no-edit --no-full-edit --no-reasoning-edit"
: > $O/r6j_configs.jsonl
run() { echo "# bench.py $*" >> $O/r6j_configs.jsonl; timeout 600 python bench.py $F "$@" 2>>$O/r6j_err.log | tail -1 >> $O/r6j_configs.jsonl; echo "rc $? $*"; }
run --guidance 1.0 --steps 16 --warmup 2
run --guidance 1.0 --steps 16 --warmup 2 --graph
run --frames 8 --steps 3 --warmup 1
run --height 1056 --width 1584 --steps 6 --warmup 1
run --height 1056 --width 1584 --steps 10 --warmup 1 --fp8
run --height 1056 --width 1584 --steps 10 --warmup 1 --fp8 --fp8-policy accurate
run --fp8 --steps 10 --warmup 2
run --fp8 --steps 10 --warmup 2 --fp8-policy accurate
python - <<'PY'
import json
for l in open("gpurun_out/r6j_configs.jsonl"):
    if l.startswith("#"):
        print(l.strip()); continue
    d = json.loads(l)
    print("  ", d["value"], d["ms_per_step"], d.get("mfma_roofline_frac_whole_step"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
PY
