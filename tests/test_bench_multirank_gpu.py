"""`bench.py --gpus N` as the driver launches it (python -m torch.distributed.run, one rank per "GPU"), on ONE GPU: with
CE_BENCH_TEST_BACKEND=gloo all ranks share cuda:0 and the collectives are host-staged, which exercises every line of the N > 1 path
(rank split, sharded step, exchange timing, sharded-vs-single verification, secondary legs, the control-plane group) without measuring
anything.  Pinned here, so that the first real 8-GPU run cannot be the first run of this code:
  * W = 2 (guidance pair split over the two ranks), 4 and 8 (one Ulysses group, pair batched inside it) print ONE JSON line that carries
    `sharded_vs_single_rel_l2` (the sharded step's latents against the unsharded step on the same inputs), the per-exchange timings
    `rccl.exchange_us_per_layer` (k|v, q, output), the replica figure and the strong-scaling reference;
  * the bare command `python bench.py --gpus 2` (no launcher) re-executes itself under torch.distributed.run and prints the same lines;
  * the complete line carries `cpu_baseline` (rank 0; the reference's own block class where oracle/_ref/transformer_ref.bin was built) and
    `sec_per_edit_temporal_reasoning` measured with the DiT sharded; the headline is also printed in a `preliminary` line before those legs;
  * an exception in the sharded leg (injected on one rank, and on all) still ends in parsable output: error lines (the failing rank's own,
    at once, and rank 0's), then the replica line of the same run.
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


def _run(world, extra=(), env_extra=None, timeout=900, bare=False):
    env = dict(os.environ, CE_BENCH_TEST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CE_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    env.update(env_extra or {})
    bench_args = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
                  "--layers", "2", "--height", "352", "--width", "640", "--no-profile", *extra]
    if bare:  # `python bench.py --gpus N`, no launcher: bench.py re-executes itself under torch.distributed.run
        cmd = [sys.executable, *bench_args]
    else:     # the driver's form
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), *bench_args]
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
    # W = 2 is launched BARE (`python bench.py --gpus 2`: bench.py must re-execute itself under torch.distributed.run - VERDICT r4 #1a) and runs
    # every leg (encoders in the sharded edit, the CPU baseline); W = 4 / 8 use the driver's launcher form and trim the legs that only cost time
    lines = _run(world, bare=(world == 2), extra=() if world == 2 else ("--no-encoders", "--no-cpu-baseline"))
    assert len(lines) == 2, lines  # the preliminary headline (printed before the minutes-long secondary legs), then the complete line
    pre, o = lines
    assert pre.get("preliminary") is True and pre["n_gpus"] == world and pre["value"] == o["value"] and pre["scaling"] == "strong"
    assert "preliminary" not in o
    # sec/edit with the DiT sharded (configs[3]: 8 latent frames, truncated to 2 after num_temporal_reasoning_steps; 3-step schedule under the test backend)
    ed = o["sec_per_edit_temporal_reasoning"]
    assert "error" not in ed and len(ed) == 2, ed
    for name, e in ed.items():
        assert e["seconds"] > 0 and e["finite"] is True and e["n_gpus"] == world and "replicated" in e["sharding"], (name, e)
    assert sorted(e["frames"] for e in ed.values()) == [5, 29]  # truncated after step 1 -> 5 pixel frames; never truncated -> 29
    if world == 2:
        cb = o["cpu_baseline"]
        assert "error" not in cb and cb["value"] > 0 and cb["cores"] >= 1, cb
        assert cb["kind"] == ("reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "transformer_ref.bin")) else "port"), cb
        assert cb["port"]["seconds_per_block"] > 0
    else:
        assert "cpu_baseline" not in o
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


def test_fp8_flags_under_sharding_say_what_ran():
    """`bench.py --gpus N --fp8` - the fp8 GEMMs shard with the rows, the MXFP8 self-attention does not exist for the sharded path
    (the exchange carries bf16 q / k / v: transformer.attention_path()), so the line's `dtype` must say bf16 attention, not claim fp8 attention;
    and the model's prediction block carries the expected wall time of the default command (6a)."""
    lines = _run(4, extra=("--fp8", "--no-secondary", "--no-cpu-baseline", "--no-encoders", "--no-reasoning-edit"))
    o = lines[-1]
    assert o["n_gpus"] == 4 and o["finite"] is True and o["value"] > 0
    assert o["dtype"].startswith("fp8 e4m3 GEMMs") and "bf16 attention" in o["dtype"] and "MXFP8" not in o["dtype"], o["dtype"]
    assert o["config"]["parallelism"].startswith("ulysses sp4")
    assert o["sharded_vs_single_rel_l2"] is None or o["sharded_vs_single_rel_l2"] < 5e-2  # (fp8 GEMMs on both sides; --no-secondary: no comparison)


@pytest.mark.parametrize("who", ["1", "all"])
def test_sharded_failure_still_prints_the_replica_line(who):
    # (one rank failing alone leaves its peers inside a collective: they run into the data group's timeout, shortened here, and the
    # host-side vote then sends every rank down the replica path; under RCCL a watchdog timeout aborts the process instead - only
    # failures that every rank sees, e.g. an RCCL initialisation or a shape error, are recoverable there)
    lines = _run(4, extra=("--no-secondary", "--no-cpu-baseline"), env_extra={"CE_BENCH_INJECT_SHARDED_FAILURE": who, "CE_BENCH_PG_TIMEOUT_S": "15"})
    errs, rest = [l for l in lines if "error" in l], [l for l in lines if "error" not in l]
    assert errs and all(e["value"] is None and e["n_gpus"] == 4 for e in errs), lines  # rank 0's line, and the failing rank's own (said at once)
    assert any("rank" not in e for e in errs), errs
    if who == "1":
        assert any(e.get("rank") == 1 for e in errs), errs
    assert len(rest) == 2 and rest[0].get("preliminary") is True, lines
    rep = rest[-1]
    assert rep["value"] > 0 and rep["scaling"] == "weak" and rep["n_gpus"] == 4 and rep["finite"] is True
    assert "sharded_error" in rep and "fallback" in rep
    assert rep["config"]["parallelism"] == "replica x4"


@pytest.mark.parametrize("graph", [False, True])
def test_one_rank_rccl_owned_comm_flags(graph):
    """`bench.py --owned-comm [--graph]` on REAL RCCL (VERDICT r4 #5a): CE_BENCH_ONE_RANK_SP puts the sharded code path of bench.py on a Ulysses group of
    one rank (every exchange a real ncclSend / ncclRecv batch on the library-owned communicator), eager and as the captured step the flag
    pair selects.  Pins the flag combination the first multi-GPU run would use; measures nothing."""
    env = dict(os.environ, CE_BENCH_ONE_RANK_SP="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CE_BENCH_TEST_BACKEND", "CE_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--layers", "2", "--height", "352", "--width", "640",
           "--no-profile", "--no-secondary", "--no-cpu-baseline", "--no-reasoning-edit", "--owned-comm", *(["--graph"] if graph else [])]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"stdout tail: {r.stdout[-1500:]}\nstderr tail: {r.stderr[-3000:]}"
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{") and ln.strip().endswith("}")]
    assert len(lines) == 1, lines
    o = lines[0]
    assert o["n_gpus"] == 1 and o["finite"] is True and "ONE_RANK_SP" in o["TEST_ONLY"] and o["value"] > 0
    assert o["launch"] == ("hipGraph replay" if graph else "eager")
    rc = o["rccl"]
    assert rc["backend"] == "nccl" and rc["world"] == 1 and rc["communicator"].startswith("library-owned"), rc
    if not graph:  # (a replay runs no Python: the call counter only sees the capture)
        assert rc["all_to_all_per_layer_per_forward"] == 3  # k|v, q, output - also with one rank (force=True)
    assert o["config"]["tokens"] == 8 * 22 * 40 and o["config"]["parallelism"].startswith("ulysses sp1")
    assert o["sharded_verification"]["latents_abs_sum_spread_over_ranks"] == 0.0
