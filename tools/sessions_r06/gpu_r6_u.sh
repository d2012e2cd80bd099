#!/bin/bash
# round 6, session u: conv -> RMS_norm -> SiLU as one launch in the 96-channel ResidualBlocks: parity, the VAE goldens, timing A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $O/r6u_pytest.txt
for f in 0 1 0 1; do
  echo "== CE_VAE_FUSE_NORM=$f" | tee -a $O/r6u_vae_bench.txt
  CE_VAE_FUSE_NORM=$f timeout 600 python tools/vae_bench.py 2>&1 | grep "hipGraph replay" | tee -a $O/r6u_vae_bench.txt
done
