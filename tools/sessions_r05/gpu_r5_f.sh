#!/bin/bash
# round 5, session f: the 256 x 96 macro tile for the VAE's 96-channel convolutions (tests, A/B, VAE timings) + the one-rank RCCL bench tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_vae_gpu.py "tests/test_bench_multirank_gpu.py::test_one_rank_rccl_owned_comm_flags" -q -x --durations=5 > gpurun_out/r5f_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5f_pytest.log )
tail -15 gpurun_out/r5f_pytest.log
timeout 600 python tools/conv_gemm_ab.py 3 > gpurun_out/r5f_conv_gemm_ab.txt 2>&1; cat gpurun_out/r5f_conv_gemm_ab.txt
timeout 600 python tools/vae_bench.py 720 1280 2 > gpurun_out/r5f_vae_bench_720p.json 2> gpurun_out/r5f_vae_bench.err; tail -c 1500 gpurun_out/r5f_vae_bench_720p.json
