#!/bin/bash
# round-2 session G: the N > 1 code path of bench.py on a one-GPU box (ranks share GPU 0, gloo host-staged collectives; 4 of 40 blocks)
mkdir -p gpurun_out
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # tag nproc extra-args...
  tag=$1; n=$2; shift 2
  CE_BENCH_TEST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
    bench.py --gpus $n --steps 1 --warmup 1 --layers 4 "$@" > gpurun_out/bench_test_$tag.log 2>&1
  echo "$tag exit $?"
  tail -1 gpurun_out/bench_test_$tag.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline())
    print({k: d.get(k) for k in ('value','n_gpus','scaling','finite','TEST_ONLY','single_gpu_same_workload_steps_per_sec','strong_scaling_speedup_vs_one_gpu','replica_mode')})
    print(d['config']); print(d['rccl'])
except Exception as e:
    print('unparsable', e)
"
  grep -E "Error|error|Traceback" gpurun_out/bench_test_$tag.log | head -5
}
run w2_default 2
run w2_ulysses 2 --no-cfg-parallel
run w4_default 4
run w4_cfgp 4 --cfg-parallel
