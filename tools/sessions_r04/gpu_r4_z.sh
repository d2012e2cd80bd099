#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z_prof -o fp8 -- python $R/bench.py --fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-edit --no-full-edit --no-reasoning-edit > $R/gpurun_out/r4z_rocprof.log 2>&1)
find gpurun_out/r4z_prof -name "*kernel_stats.csv" -exec head -14 {} \; | cut -c1-160
