#!/bin/bash
# round 6, session x: the 288 x 256 macro tile (NF = 9 form of ce_gemm384.hip): parity of variant 7, all three tiles timed at every row count
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "tile288x256 or tile384x256" 2>&1 | tail -4 | tee $O/r6x_pytest.txt
timeout 1500 python tools/gemm_tile_choice.py 7200,14400,13068,26136,28800,3648,7296 3 2>&1 | grep -v amdgpu.ids | tee $O/r6x_gemm_tile_choice.txt
