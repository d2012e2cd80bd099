#!/bin/bash
# round-4 GPU session C: PMC passes of both bodies of the V^T attention kernel (7200 x 40 heads x 2 samples = the step's launch; 28800 keys)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for body in 0 128; do
  CE_ATTN_WAVES=$body bash tools/gpu_pmc.sh attnvt_7200_b2_body$body attnvt 7200 40 2 3 > gpurun_out/r4c_pmc_attnvt_7200_b2_body$body.txt 2>&1
  CE_ATTN_WAVES=$body bash tools/gpu_pmc.sh attnvt_28800_body$body attnvt 28800 40 1 2 > gpurun_out/r4c_pmc_attnvt_28800_body$body.txt 2>&1
done
tail -n 40 gpurun_out/r4c_pmc_attnvt_7200_b2_body0.txt gpurun_out/r4c_pmc_attnvt_7200_b2_body128.txt
