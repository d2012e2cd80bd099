#!/bin/bash
# round 6, session c: the refitted macro-tile choice (ce_gemm_bf16_tile_rows) against both tiles measured at every row count of the engine
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 1500 python tools/gemm_tile_choice.py 13068,26136,28800,57600,3648,7296 2>&1 | grep -v amdgpu.ids | tee $O/r6c_gemm_tile_choice_2.txt
