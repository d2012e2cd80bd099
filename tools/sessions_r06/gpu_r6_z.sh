#!/bin/bash
# round 6, session z: final evidence on the final tree - the whole GPU suite (timed), smoke, the VAE account + kernel stats, the driver's bench command,
# complete lines for the other BASELINE configs, rocprofv3 kernel stats of configs[2] (the 288-row tile is in its step now)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out; O=gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=25 -x ) > $O/r6z_pytest.log 2>&1; tail -8 $O/r6z_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
CE_VAE_BENCH_OUT=$O/r6z_vae_bench_720p.json timeout 600 python tools/vae_bench.py 2>&1 | grep "hipGraph replay"
(cd /tmp && CE_VAE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6z_prof_vae -o vae -- python $R/tools/vae_bench.py > $R/$O/r6z_rocprof_vae.log 2>&1)
f=$(find $O/r6z_prof_vae -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r6z_vae_kernel_stats.csv; rm -rf $O/r6z_prof_vae
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6z_bench.json 2> $O/r6z_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6z_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['mfma_roofline_frac_whole_step'], d['steps_per_sec_fp8_mode'], d['fp8_mode_frac_of_fp8_peak'], d['steps_per_sec_fp8_config4']['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_family']['frac'])
print({k: v['seconds'] for k, v in d['sec_per_edit'].items() if isinstance(v, dict)}, d['vae'], d['power'])
PY
F="--no-vae --no-encoders --no-fp8-leg --no-fp8-config4 --no-cpu-baseline --no-edit --no-full-edit --no-reasoning-edit"
: > $O/r6z_configs.jsonl
run() { echo "# bench.py $*" >> $O/r6z_configs.jsonl; timeout 600 python bench.py $F "$@" 2>>$O/r6z_err.log | tail -1 >> $O/r6z_configs.jsonl; echo "rc $? $*"; }
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
for l in open("gpurun_out/r6z_configs.jsonl"):
    if l.startswith("#"):
        print(l.strip()); continue
    d = json.loads(l)
    print("  ", d["value"], d["ms_per_step"], d.get("mfma_roofline_frac_whole_step"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
PY
F2="--no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg --no-edit --no-full-edit --no-reasoning-edit"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6z_prof_c2 -o bench -- python $R/bench.py --steps 5 --warmup 1 $F2 --guidance 1.0 --steps 8 > $R/$O/r6z_rocprof_c2.log 2>&1)
f=$(find $O/r6z_prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r6z_prof_c2_kernel_stats.csv; rm -rf $O/r6z_prof_c2
head -8 $O/r6z_prof_c2_kernel_stats.csv | cut -c1-160
