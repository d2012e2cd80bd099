#!/bin/bash
# round 6, session q: the whole -m gpu suite with the full-width oracle evaluations on the device and the new VAE attention; every duration
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=100 -x ) > $O/r6q_pytest.log 2>&1; tail -8 $O/r6q_pytest.log
