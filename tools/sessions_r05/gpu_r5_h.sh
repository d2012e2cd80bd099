#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/attn_overhead_probe.py > gpurun_out/r5h_attn_overhead_probe.txt 2>&1; cat gpurun_out/r5h_attn_overhead_probe.txt
