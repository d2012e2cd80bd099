#!/bin/bash
# round-2 session N: rocprofv3 kernel stats of the fp8-mode bench (1584x1056) and of the N = 28 800 one-GPU line
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fp8 -o bench -- python $R/bench.py --fp8 --height 1056 --width 1584 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-edit > $R/gpurun_out/rocprof_fp8.log 2>&1)
head -12 gpurun_out/prof_fp8/bench_kernel_stats.csv | cut -c1-150
