#!/bin/bash
# round 6, session ac: the step with V^T by the transposed store vs by the swapped product (CE_VT_GEMM=row), one box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
F="--no-vae --no-encoders --no-fp8-leg --no-fp8-config4 --no-cpu-baseline --no-edit --no-full-edit --no-reasoning-edit --no-profile"
: > $O/r6ac_vt_step_ab.txt
for rep in 1 2 3; do
  for v in row T; do
    for a in "--steps 12 --warmup 3" "--guidance 1.0 --steps 20 --warmup 3"; do
      r=$(CE_VT_GEMM=$v timeout 600 python bench.py $F $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
      echo "rep $rep V^T=$v [$a]: $r" | tee -a $O/r6ac_vt_step_ab.txt
    done
  done
done
