#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vae_gpu.py -q --no-header -p no:cacheprovider -x -k "720p or rms_silu or graph" > gpurun_out/t_pytest_vae.log 2>&1; grep -v amdgpu.ids gpurun_out/t_pytest_vae.log | tail -8
