#!/bin/bash
# Multi-GPU scaling curve on ONE node with N MI355X (the driver runs the same commands at round end):
#   tools/scale_bench.sh [steps] [warmup]      ->  gpurun_out/scale_n{1,2,4,8}.json
# N = 1: configs[1] (7200 tokens).  N > 1: configs[3] (28 800 tokens), ONE edit sharded Ulysses-style over the N GPUs with
# RCCL all-to-all over xGMI; each line also carries the one-GPU rate of the same workload and the replica (weak) figure.
steps=${1:-4}; warm=${2:-1}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps $steps --warmup $warm | tail -1 > gpurun_out/scale_n1.json
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
port=29531
for n in 2 4 8; do
  [ "$n" -le "$ngpu" ] || continue
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py \
      --gpus $n --steps $steps --warmup $warm | tail -1 > gpurun_out/scale_n$n.json
  port=$((port+1))
  # the other split of the same N ranks (default: guidance pair split on 2 GPUs, one Ulysses group from 4 on - DESIGN.md section 6)
  alt=$([ "$n" -eq 2 ] && echo "--no-cfg-parallel" || echo "--cfg-parallel")
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py \
      --gpus $n --steps $steps --warmup $warm $alt --no-secondary | tail -1 > gpurun_out/scale_n${n}_alt.json
  port=$((port+1))
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/scale_n*.json")):
    try:
        d = json.load(open(f))
        print(f, d["n_gpus"], d["config"]["parallelism"], d["value"], d["unit"], "| 1-GPU same workload:", d.get("single_gpu_same_workload_steps_per_sec"),
              "| speed-up:", d.get("strong_scaling_speedup_vs_one_gpu"), "| replica:", (d.get("replica_mode") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
