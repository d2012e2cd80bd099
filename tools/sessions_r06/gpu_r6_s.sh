#!/bin/bash
# round 6, session s: PMC passes of the 96-channel conv - the slab kernel (n_tile 1) and the 256 x 96 GEMM tile - and of the VAE attention
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
bash tools/gpu_pmc.sh conv96_slab conv 3 96 96 4 720 1280 1 3 > $O/r6s_pmc_conv96_slab.txt 2>&1
bash tools/gpu_pmc.sh conv96_gemm conv 3 96 96 4 720 1280 96 3 > $O/r6s_pmc_conv96_gemm.txt 2>&1
bash tools/gpu_pmc.sh attn1 attn1 14400 384 3 > $O/r6s_pmc_attn1.txt 2>&1
grep -A12 "conv3x3_c96\|gemm_bf16_w4\|attn_1head_kernel" $O/r6s_pmc_conv96_slab.txt $O/r6s_pmc_conv96_gemm.txt $O/r6s_pmc_attn1.txt | grep -v "^--" | cut -c1-160
