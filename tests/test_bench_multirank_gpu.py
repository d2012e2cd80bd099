"""`bench.py --gpus N` as the driver launches it (python -m torch.distributed.run, one rank per "GPU"), on ONE GPU: with
CE_BENCH_TEST_BACKEND=gloo all ranks share cuda:0 and the collectives are host-staged, which exercises every line of the N > 1 path
(rank split, sharded step, exchange timing, sharded-vs-single verification, secondary legs, the control-plane group) without measuring
anything.  Pinned here, so that the first real 8-GPU run cannot be the first run of this code:
  * W = 2 (guidance pair split over the two ranks), 4 and 8 (one Ulysses group, pair batched inside it) print ONE JSON line that carries
    `sharded_vs_single_rel_l2` (the sharded step's latents against the unsharded step on the same inputs), the per-exchange timings
    `rccl.exchange_us_per_layer` (k|v, q, output), the replica figure and the strong-scaling reference;
  * an exception in the sharded leg (injected on one rank, and on all) still ends in parsable output: an error line, then the replica
    line of the same run.
Reduced depth (2 blocks) and resolution (352x640, 8 latent frames = 7 040 tokens): the lines are marked invalid / TEST_ONLY by bench.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra=(), env_extra=None, timeout=900):
    env = dict(os.environ, CE_BENCH_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--layers", "2", "--height", "352", "--width", "640", "--no-profile", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = []
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    assert r.returncode == 0, f"bench.py --gpus {world} exited {r.returncode}\nstdout tail: {r.stdout[-2000:]}\nstderr tail: {r.stderr[-3000:]}"
    return lines


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_line_verifies_itself(world):
    lines = _run(world)
    assert len(lines) == 1, lines
    o = lines[0]
    assert o["n_gpus"] == world and o["scaling"] == "strong" and o["finite"] is True and "TEST_ONLY" in o
    assert o["value"] > 0 and o["steps"] == 1
    # the sharded answer, checked against the unsharded step on the same inputs inside the same run
    assert o["sharded_vs_single_rel_l2"] is not None and o["sharded_vs_single_rel_l2"] < 1e-2, o["sharded_verification"]
    v = o["sharded_verification"]
    assert v["sharded_vs_single_rel_l2_of_update"] < 3e-2, v
    assert v["latents_abs_sum_spread_over_ranks"] < 1e-6, v  # every rank holds the same replicated latents
    assert o["single_gpu_same_workload_steps_per_sec"] > 0 and o["strong_scaling_speedup_vs_one_gpu"] > 0
    assert o["replica_mode"]["value"] > 0 and o["replica_mode"]["scaling"] == "weak"
    rc = o["rccl"]
    assert rc["world"] == world
    if world == 2:  # the guidance pair split: one whole forward per rank, no all-to-all inside a forward
        assert rc["cfg_parallel_groups"] == 2 and rc["ulysses_group"] == 1 and rc["exchange_us_per_layer"] is None
    else:
        assert rc["cfg_parallel_groups"] == 1 and rc["ulysses_group"] == world and rc["all_to_all_per_layer_per_forward"] == 3
        ex = rc["exchange_us_per_layer"]
        assert set(ex) == {"k|v", "q", "output"}
        for name, e in ex.items():
            assert e["us"] > 0 and e["bytes_sent_off_rank"] > 0 and e["GBps_per_rank"] > 0, (name, e)
        assert ex["k|v"]["bytes_sent_off_rank"] == 2 * ex["q"]["bytes_sent_off_rank"] == 2 * ex["output"]["bytes_sent_off_rank"]


@pytest.mark.parametrize("who", ["1", "all"])
def test_sharded_failure_still_prints_the_replica_line(who):
    # (one rank failing alone leaves its peers inside a collective: they run into the data group's timeout, shortened here, and the
    # host-side vote then sends every rank down the replica path; under RCCL a watchdog timeout aborts the process instead - only
    # failures that every rank sees, e.g. an RCCL initialisation or a shape error, are recoverable there)
    lines = _run(4, extra=("--no-secondary",), env_extra={"CE_BENCH_INJECT_SHARDED_FAILURE": who, "CE_BENCH_PG_TIMEOUT_S": "30"})
    assert len(lines) == 2, lines
    err, rep = lines
    assert err["value"] is None and "error" in err and err["n_gpus"] == 4
    assert rep["value"] > 0 and rep["scaling"] == "weak" and rep["n_gpus"] == 4 and rep["finite"] is True
    assert "sharded_error" in rep and "fallback" in rep
    assert rep["config"]["parallelism"] == "replica x4"
