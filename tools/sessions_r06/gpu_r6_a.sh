#!/bin/bash
# round 6, session a: evidence for BASELINE configs[2] / [3] / [4] on the round-5 kernels (before any change of this round):
#  * complete bench lines (kernel_breakdown + roofline; no truncation) for configs[2] eager + hipGraph, configs[3]'s shape on one GPU, configs[4] bf16 / fp8
#  * rocprofv3 --kernel-trace --stats of the configs[2] step (guidance 1, B = 1, M = 7 200) and, for comparison, of the guidance-5 step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
O=gpurun_out
F="--no-vae --no-encoders --no-fp8-leg --no-fp8-config4 --no-cpu-baseline --no-edit --no-full-edit --no-reasoning-edit"
: > $O/r6a_configs.jsonl
run() { echo "# $*" >> $O/r6a_configs.jsonl; timeout 600 python bench.py $F "$@" 2>$O/r6a_err.log | tail -1 >> $O/r6a_configs.jsonl; echo "rc $? $*"; }
run --guidance 1.0 --steps 16 --warmup 2
run --guidance 1.0 --steps 16 --warmup 2 --graph
run --frames 8 --steps 3 --warmup 1
run --height 1056 --width 1584 --steps 6 --warmup 1
run --height 1056 --width 1584 --steps 10 --warmup 1 --fp8
run --fp8 --steps 10 --warmup 2
G="$F --no-profile"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6a_prof_c2 -o bench -- python $R/bench.py --guidance 1.0 --steps 8 --warmup 1 $G > $R/$O/r6a_rocprof_c2.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r6a_prof_c1 -o bench -- python $R/bench.py --steps 5 --warmup 1 $G > $R/$O/r6a_rocprof_c1.log 2>&1)
for d in r6a_prof_c2 r6a_prof_c1; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
python - <<'PY'
import json
for l in open("gpurun_out/r6a_configs.jsonl"):
    if l.startswith("#"):
        print(l.strip()); continue
    try:
        d = json.loads(l)
        print("  ", d["value"], d["unit"], d["ms_per_step"], d.get("achieved_tflops_per_gpu"), d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
        for k, v in list((d.get("kernel_breakdown") or {}).items())[:16]:
            print("      ", k, v)
    except Exception as e:
        print("   ?", e, l[:200])
PY
head -30 $O/r6a_prof_c2_kernel_stats.csv
