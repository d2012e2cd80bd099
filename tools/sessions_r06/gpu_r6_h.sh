#!/bin/bash
# round 6, session h: what is left at the tile boundaries of the 384-row GEMM - a tile without waiting for its first K-tile, and without its epilogue
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; L=chronoedit_amd/lib
timeout 900 python tools/gemm_ab.py $L/libchronoedit_hip.so $L/libce_g384_noprologue.so $L/libce_g384_noepilogue.so 2>&1 | grep -v amdgpu.ids | tee $O/r6h_gemm384_boundary_ablation.txt
