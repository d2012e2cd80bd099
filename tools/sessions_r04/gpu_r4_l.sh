#!/bin/bash
mkdir -p gpurun_out
timeout 60 tools/probes/mx16_probe > gpurun_out/r4l_mx16_probe.txt 2>&1
cat gpurun_out/r4l_mx16_probe.txt
