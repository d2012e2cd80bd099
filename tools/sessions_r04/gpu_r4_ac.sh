#!/bin/bash
# PMC traffic / L2 passes of two GEMM shapes of the step on the round-4 build (variant -1 = the per-shape tile choice the step uses)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_pmc_traffic.sh gemm_outproj gemm 14400 5120 5120 2 -1 3 > /dev/null 2>&1
bash tools/gpu_pmc_traffic.sh gemm_ffnup gemm 14400 13824 5120 1 -1 3 > /dev/null 2>&1
grep -v "at::\|transpose" gpurun_out/pmc_gemm_outproj.txt | head -40
echo ======
grep -v "at::\|transpose" gpurun_out/pmc_gemm_ffnup.txt | head -40
