#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu_pmc.sh attn_sp3 attn 7200 7200 40 3 > gpurun_out/pmc_attn_sp3.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-vae --no-encoders --no-fp8-leg > $R/gpurun_out/rocprof.log 2>&1)
timeout 500 python tools/microbench.py attn gemm row 2>&1 | grep -v amdgpu.ids > gpurun_out/microbench.log
timeout 300 python tools/full_edit.py --steps 50 2>/dev/null | tail -1 > gpurun_out/full_edit_50.json
timeout 200 python tools/full_edit.py --steps 8 --guidance 1.0 2>/dev/null | tail -1 > gpurun_out/full_edit_8.json
tail -3 gpurun_out/microbench.log; cat gpurun_out/full_edit_50.json | cut -c1-300; ls gpurun_out/prof | head
