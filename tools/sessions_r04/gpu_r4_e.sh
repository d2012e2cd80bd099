#!/bin/bash
mkdir -p gpurun_out
CE_ATTN_WAVES=128 bash tools/gpu_pmc.sh attnvt_28800_body128_v2 attnvt 28800 40 1 2 > gpurun_out/r4e_pmc_attnvt_28800_body128_v2.txt 2>&1
grep -A12 "attn_fwd" gpurun_out/r4e_pmc_attnvt_28800_body128_v2.txt | grep -v "^--" | awk '{$1=$1};1' | sort -u | grep -v csv
