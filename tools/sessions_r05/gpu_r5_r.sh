#!/bin/bash
# round 5, session r: the fp8 GEMM with an A ring of three stages (F8_A3=1) against the two-stage loop - stand-alone (cold weights) and inside the fp8 step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=chronoedit_amd/lib
timeout 900 python tools/gemm_mxfp8_ab.py --cold $L/libce_a3_0.so $L/libce_a3_1.so > gpurun_out/r5r_gemm_mxfp8_a3_ab.txt 2>&1
cat gpurun_out/r5r_gemm_mxfp8_a3_ab.txt
F="--fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-vae --no-encoders --no-fp8-leg --no-edit --no-profile"
for rep in 1 2; do
for v in 0 1; do
  r=$(CE_HIPLIB_PATH=$PWD/$L/libce_a3_$v.so timeout 300 python bench.py $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['finite'])")
  echo "rep $rep  F8_A3=$v  720p fp8 step: steps/s, ms/step, finite = $r" | tee -a gpurun_out/r5r_gemm_mxfp8_a3_ab.txt
done
done
