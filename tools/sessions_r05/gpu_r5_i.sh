#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/probes/attn_shape_probe > gpurun_out/r5i_attn_shape_probe.txt 2>&1; cat gpurun_out/r5i_attn_shape_probe.txt
