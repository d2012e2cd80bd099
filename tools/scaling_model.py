"""Strong-scaling PREDICTION for `bench.py --gpus N` (Ulysses over RCCL, BASELINE.json configs[3]: N = 28 800 tokens, guidance 5).

A model, not a measurement: it prices the sharded step from (a) the per-kernel times measured on ONE MI355X for the same workload
(profiles/r02_bench_n28800_one_gpu.json), (b) the tile / workgroup-round quantisation the smaller per-rank shapes meet on 256 CUs
and (c) the xGMI all-to-all time of the three exchanges per layer.  DESIGN.md section 6 quotes its output next to what the
driver's SCALE run measured.  Usage: python tools/scaling_model.py [--link-GBps 55] [--latency-us 25]"""
import argparse
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOK, D, F, H, L, CUS = 28800, 5120, 13824, 40, 40, 256


def rounds(n_wg: float) -> float:
    """Workgroups on 256 CUs, one per CU: whole rounds, except that a thin last round is worth its fill (the split-K tail of the
    GEMM and the remainder-last order of the attention spread it) - priced at max(fill, 1/2) of a round."""
    full, frac = divmod(n_wg / CUS, 1.0)
    return full + (0.0 if frac < 1e-9 else max(frac, 0.5))


def model(W: int, cfg_parallel: bool, link_GBps: float, latency_us: float, one_gpu: dict, pair_batched: bool = False, captured: bool = False):
    """captured: the sharded step replayed as a hipGraph on the library-owned communicator - every exchange then sits on the capturing stream
    (parallel.OwnedComm.all_to_all), so the k|v exchange no longer hides behind the q projection: all three exchanges are exposed."""
    sp = W // 2 if cfg_parallel else W  # ranks per Ulysses group
    # samples per forward: 1 (the two guidance passes in sequence, or side by side on two groups: cfg-parallel) or 2 (pair_batched: one
    # sharded forward of B = 2 on the blocked-layout kernels; shards rounded up to 64 tokens)
    batch = 2 if (pair_batched and not cfg_parallel) else 1
    n_loc = math.ceil(N_TOK / sp)
    if batch == 2:
        n_loc = (n_loc + 63) // 64 * 64
    rows = batch * n_loc
    heads = H // sp
    kb = one_gpu["kernel_breakdown"]

    # ---- attention: measured 26.47 ms for 2 samples x 40 heads x 113 query blocks on one GPU = per (workgroup round) cost
    wg1 = 2 * H * math.ceil(N_TOK / 256)
    per_round = kb["attention_28800x28800+0_h40_b2"]["avg_ms"] / rounds(wg1)
    attn = per_round * rounds(batch * heads * math.ceil(N_TOK / 256))
    xattn = kb["attention_28800x512+257_h40_b2"]["avg_ms"] / sp  # cross-attention: queries sharded by token, all heads local

    # ---- GEMMs: measured per launch at M = 57 600 rows; tiles of 256 x 256, rounds of 256 CUs
    def gemm(key, ncols):
        t1 = kb[key]["avg_ms"]
        per = t1 / rounds(math.ceil(57600 / 256) * (ncols // 256))
        return per * rounds(math.ceil(rows / 256) * (ncols // 256))
    # (the one-GPU line projects q | k as one GEMM and V^T as the swapped one; the sharded path runs [k | v] and q: same column counts)
    g = (gemm("gemm_57600x10240x5120_epi0", 10240) + gemm("gemm_57600x5120x5120_epi0", 5120) + gemm("gemm_57600x13824x5120_epi1", 13824)
         + gemm("gemm_57600x5120x13824_epi2", 5120) + 2 * gemm("gemm_57600x5120x5120_epi2", 5120) + gemm("gemm_57600x5120x5120_epi0", 5120))
    q_gemm = gemm("gemm_57600x5120x5120_epi0", 5120)  # what the k|v exchange hides behind (the q third of the fused projection)
    row = (3 * kb["ln_affine_57600x5120"]["avg_ms"] + kb["rmsnorm_rope_57600x5120x2"]["avg_ms"] + kb["rmsnorm_rope_57600x5120"]["avg_ms"]) * rows / 57600

    # ---- exchanges (per layer, per rank): k|v, q, attention output; bytes leaving a rank over EACH of its sp-1 links
    def a2a_ms(ntensors):
        if sp == 1:
            return 0.0
        per_link = ntensors * rows * (D // sp) * 2  # [local rows][D / sp] bf16 to every peer
        return per_link / (link_GBps * 1e9) * 1e3 + latency_us * 1e-3
    kv, q, o = a2a_ms(2), a2a_ms(1), a2a_ms(1)
    exposed = (kv if captured else max(0.0, kv - q_gemm)) + q + o
    layer = attn + xattn + g + row + exposed
    fwd = L * layer + 6.0 / sp  # + context K/V projections (3 + 3 ms on one GPU), head, patchify
    passes = 1 if (cfg_parallel or batch == 2) else 2
    step = passes * fwd + (0.1 if cfg_parallel else 0.0)  # + the 3.7 MB prediction exchange
    return dict(W=W, mode="cfg-parallel 2 x %d" % sp if cfg_parallel else ("ulysses %d, B=2" % sp if batch == 2 else "ulysses %d" % sp), rows=rows, heads=heads,
                attn_ms=attn, gemm_ms=g, exchange_ms=kv + q + o, exposed_ms=exposed, step_ms=step, steps_per_s=1e3 / step)


def predict(W: int, link_GBps: float = 55.0, latency_us: float = 25.0) -> dict:
    """What `bench.py --gpus W` puts into its `rccl` block next to the measurement (full-size configs[3] only): the model's steps/s for every
    way of splitting W ranks, eager and captured, so that the first real multi-GPU run can be read against it.  `choice` = the split and
    the launch form the model ranks first - a captured loop only wins where the host cannot keep W ranks' queues full, which the one-GPU
    enqueue time (8 ... 16 ms per step against >= 250 ms of GPU work per rank at W = 8) says is never the case at these shapes."""
    one = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_n28800_one_gpu.json")))
    out = {"source": "tools/scaling_model.py on profiles/r02_bench_n28800_one_gpu.json", "link_GBps_one_way": link_GBps, "latency_us_per_collective": latency_us,
           "one_gpu_steps_per_s": round(1e3 / one["ms_per_step"], 4), "steps_per_s": {}}
    for cfgp, pb in ((False, False), (False, True), (True, False)):
        if cfgp and (W < 2 or W % 2):
            continue
        for cap in (False, True):
            if cap and (cfgp or W == 1):
                continue  # the guidance-pair exchange of CFG parallelism is a torch.distributed collective: never captured
            r = model(W, cfgp, link_GBps, latency_us, one, pair_batched=pb, captured=cap)
            out["steps_per_s"][r["mode"] + (", hipGraph" if cap else ", eager")] = round(r["steps_per_s"], 4)
    best = max(out["steps_per_s"], key=out["steps_per_s"].get)
    out["choice"] = best
    out["expected_driver_wall_s"] = expected_wall_s(W, 1e3 / out["steps_per_s"][best])
    return out


def expected_wall_s(W: int, step_ms: float, steps: int = 20, warmup: int = 5, reasoning_steps=(10, 50)) -> dict:
    """How long the DEFAULT `bench.py --gpus W --steps 20 --warmup 5` line should take end to end on W real GPUs: the
    legs of bench.py priced with the model's sharded step time and the one-GPU measurements of profiles/r05_bench.json - so that a first
    multi-GPU run that exceeds it by more than ~2x can be called a hang and not a slow run.  Seconds; an estimate (+-30 %)."""
    sp_up = 2103.0 / step_ms                      # the model's strong-scaling speed-up at 8 latent frames
    step2_ms = max(339.0 / (0.8 * sp_up), 45.0)   # the 2-frame step (N = 7 200) sharded: ~80 % of that speed-up (smaller per-rank shapes)
    legs = {
        "import torch + model build (40 blocks, 30.5 GiB of random weights per rank) + process-group init": 60.0,
        "timed region: (warmup + steps) sharded steps": (warmup + steps) * step_ms / 1e3,
        "profile step + sharded-vs-single verification (one sharded step, one unsharded on rank 0) + exchange timing": 2 * step_ms / 1e3 + 2.2 + 1.0,
        "secondaries on rank 0: 4 unsharded steps at 8 frames + 3 replica steps at 2 frames": 4 * 2.1 + 3 * 0.34,
        "sharded temporal-reasoning edits (VAE of 29 frames + two decodes + encoders replicated: ~1.5 s each)":
            sum(rs * step_ms / 1e3 + (50 - rs) * step2_ms / 1e3 + 1.5 for rs in reasoning_steps) + 12.0,  # + encoder / VAE weight build
        "cpu_baseline on rank 0 (the reference's block at N = 28 800 on the host cores, one bounded run) + cpu_config0": 110.0,
    }
    return {"total": round(sum(legs.values()), 0), "legs": {k: round(v, 1) for k, v in legs.items()},
            "note": "estimate from tools/scaling_model.py (+-30 %); one GPU measured 224 s for its (different) set of legs in round 5"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--link-GBps", type=float, default=55.0, help="sustained one-direction rate of one xGMI link inside an RCCL all-to-all "
                    "(153.6 GB/s per link both directions -> 76.8 one way at the data sheet; ~70 % of it assumed)")
    ap.add_argument("--latency-us", type=float, default=25.0, help="fixed cost of one all_to_all_single (launch + sync)")
    a = ap.parse_args()
    one = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_n28800_one_gpu.json")))
    base = one["ms_per_step"]
    print(f"measured on one MI355X: {base:.0f} ms/step ({1e3 / base:.3f} steps/s); link {a.link_GBps} GB/s one way, {a.latency_us} us per collective")
    print(f"{'GPUs':>4} {'mode (captured loop: steps/s)':>34} {'rows/rank':>9} {'heads':>5} {'attn':>7} {'GEMMs':>7} {'a2a':>6} {'exposed':>7} | {'ms/step':>8} {'steps/s':>8} {'speed-up':>8} {'eff.':>5}")
    for W in (1, 2, 4, 8):
        for cfgp, pb in (((False, False), (False, True)) if W == 1 else ((False, False), (False, True), (True, False))):
            r = model(W, cfgp, a.link_GBps, a.latency_us, one, pair_batched=pb)
            if W > 1 and not cfgp:
                rc = model(W, cfgp, a.link_GBps, a.latency_us, one, pair_batched=pb, captured=True)
                r["mode"] += " (graph: %.3f)" % rc["steps_per_s"]
            print(f"{W:>4} {r['mode']:>34} {r['rows']:>9} {r['heads']:>5} {r['attn_ms']:>7.2f} {r['gemm_ms']:>7.2f} {r['exchange_ms']:>6.2f} {r['exposed_ms']:>7.2f} |"
                  f" {r['step_ms']:>8.1f} {r['steps_per_s']:>8.3f} {base / r['step_ms']:>8.2f} {base / r['step_ms'] / W:>5.2f}")


if __name__ == "__main__":
    main()
