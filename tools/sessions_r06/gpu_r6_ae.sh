#!/bin/bash
# round 6, session ae: PMC passes of the V^T product - transposed store (epi 7) against the same product stored untransposed (epi 0) on the 384-row tile
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
bash tools/gpu_pmc.sh vt_T gemm 14400 5120 5120 7 -1 4 > $O/r6ae_pmc_vt_transposed.txt 2>&1
bash tools/gpu_pmc.sh vt_plain gemm 14400 5120 5120 0 -1 4 > $O/r6ae_pmc_vt_plain.txt 2>&1
for f in $O/r6ae_pmc_vt_transposed.txt $O/r6ae_pmc_vt_plain.txt; do echo "== $f"; grep -A9 "gemm_bf16_384" $f | grep -v "^--" | grep "mean\|gemm_bf16" | cut -c1-130; done
