#!/bin/bash
# round 6, session aa: PMC passes of the slab conv kernel in its final form (two waves per SIMD, XCD-contiguous tile order)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
bash tools/gpu_pmc.sh conv96_slab_w8 conv 3 96 96 4 720 1280 1 3 > $O/r6aa_pmc_conv96_slab_w8.txt 2>&1
grep -A9 "conv3x3_c96" $O/r6aa_pmc_conv96_slab_w8.txt | grep -v "^--" | cut -c1-140
