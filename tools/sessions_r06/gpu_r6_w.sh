#!/bin/bash
# round 6, session w: find the crash of session u (two-output conv epilogue)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 300 python -m pytest tests/test_vae_gpu.py -x -q -m gpu -k "next_norm" > $O/r6w_a.log 2>&1; echo "rc $?"; grep -v "^  File\|Extension modules" $O/r6w_a.log | tail -30
timeout 300 python -m pytest tests/test_vae_gpu.py -x -q -m gpu > $O/r6w_b.log 2>&1; echo "rc $?"; grep -v "^  File\|Extension modules" $O/r6w_b.log | tail -30
