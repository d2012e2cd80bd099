#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp CE_VAE_GRAPH=0
R=$GRAFT_REPO_ROOT
(cd /tmp && CE_VAE_BENCH_OUT=$R/gpurun_out/r4ab_vae.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4ab_prof -o vae -- python $R/tools/vae_bench.py 720 1280 5 > $R/gpurun_out/r4ab_rocprof.log 2>&1)
find gpurun_out/r4ab_prof -name "*kernel_stats.csv" -exec head -16 {} \; | cut -c1-200
