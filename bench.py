#!/usr/bin/env python
"""bench.py — denoising-steps/sec of the ChronoEdit-14B hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one iteration of ChronoEditPipeline.__call__'s loop
(/root/reference/chronoedit_diffusers/pipeline_chronoedit.py:695-756) at BASELINE.json configs[1]:
ChronoEdit-14B bf16, 1280x720, 5 pixel frames -> latents [1,16,2,90,160] -> N = 7200 tokens,
guidance 5.0 -> TWO DiT forwards + CFG + flow-UniPC update.  Synthetic (seeded) weights of the real
architecture and synthetic inputs — there are no checkpoints offline.  Inputs are resident in HBM
before the timed region; nothing is cached across steps unless --cache-context is given (then the
step-invariant text/image K/V are reused, and the JSON says so).

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): each rank denoises its own
edit (independent replicas, no data-path collective; "scaling": "weak").  Timing: barrier +
synchronize on both sides, MAX over ranks.

The JSON line carries:
  roofline      dominant kernel (FFN-up GEMM) algorithmic FLOPs / mean launch duration measured with HIP events
                on the launch stream in one extra profiled step after the timed region, vs 2.5 PFLOP/s dense bf16;
  cpu_baseline  the CPU oracle (oracle/dit_oracle.py, "port") timed on this host's cores on ONE transformer block
                at the same N (rank 0, N=1 only), extrapolated to steps/sec.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=40, help="(debug only) fewer blocks => result marked invalid")
    ap.add_argument("--frames", type=int, default=2, help="latent frames: 2 (edit) or 8 (temporal reasoning)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--cache-context", action="store_true", help="reuse step-invariant text/image K/V across steps")
    ap.add_argument("--sequential-cfg", action="store_true", help="two B=1 forwards per step instead of one batched B=2 forward")
    ap.add_argument("--parallel", choices=["replica", "ulysses"], default="replica",
                    help="N>1: independent edits per GPU (weak scaling, default) or ONE edit with the token axis sharded "
                         "over the GPUs (Ulysses all-to-all over RCCL/xGMI, strong scaling)")
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph-captured step instead of launching eagerly")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE encode/decode timing used for the sec/edit figure")
    ap.add_argument("--no-encoders", action="store_true", help="skip the UMT5 / CLIP timing used for the sec/edit figure")
    ap.add_argument("--no-fp8-leg", action="store_true", help="skip the secondary fp8-GEMM-mode timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE.json configs[4] arithmetic: the six large Linears of every block in fp8 e4m3 (MX matrix instruction); "
                         "reported with dtype fp8, never the headline bf16 number")
    ap.add_argument("--attn-kernel", type=int, default=0,
                    help="(tuning) self-attention kernel knob of ce_set_attention_waves: 0 auto, 32 ping-pong, 64 sw-pipelined, 128 w4")
    return ap.parse_args()


def build_model(layers: int, dev):
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    m = ChronoEditTransformer3DModel(num_attention_heads=40, attention_head_dim=128, in_channels=36, out_channels=16,
                                     text_dim=4096, freq_dim=256, ffn_dim=13824, num_layers=layers, image_dim=1280,
                                     added_kv_proj_dim=5120, device=dev, dtype=torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(1234)
    D = 5120
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("scale_shift_table"):
                p.copy_(torch.randn(p.shape, generator=g, device=dev) / D**0.5)
            elif "norm" in name and name.endswith(".weight"):
                p.fill_(1.0)
            elif name.endswith(".bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)
    return m


def cpu_baseline(N: int, steps_fwd: int):
    """Oracle ("port") on the host cores: one full-width transformer block at N tokens, fp32."""
    from oracle import dit_oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    cfg = O.DiTConfig(num_layers=1)
    g = torch.Generator().manual_seed(1)
    p = {k: v for k, v in O.make_synthetic_params(cfg, seed=1).items() if k.startswith("blocks.0.")}
    x = torch.randn(1, N, cfg.inner_dim, generator=g)
    enc = torch.randn(1, 769, cfg.inner_dim, generator=g)
    temb6 = torch.randn(1, 6, cfg.inner_dim, generator=g) * 0.1
    T, hp, wp = 2, 45, N // 90
    rot = O.rope_table(cfg, T, 2 * hp, 2 * wp) if T * hp * wp == N else None
    with torch.no_grad():
        t0 = time.perf_counter()
        O.block_forward(p, 0, cfg, x, enc, temb6, rot)
        dt = time.perf_counter() - t0
    per_step = dt * 40 * steps_fwd
    return {"value": 1.0 / per_step, "unit": "denoising-steps/sec", "cores": cores, "kind": "port",
            "sample": f"1 of 40 DiT blocks, N={N}, fp32 torch-CPU oracle, {dt:.2f} s measured; x40 blocks x{steps_fwd} forwards/step"}


def _pmc_traffic(kernel_label: str):
    """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE
    are collected in their own runs, tools/gpu_pmc.sh; they cannot be read live from inside this process).  None when the
    shape of this run has no committed measurement."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(kernel_label)
    except (OSError, ValueError):
        rec = None
    if not rec:
        return None, None
    return rec["fetch_bytes"] + rec["write_bytes"], "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"


def _baseline_config_name(a, T) -> str:
    """Which BASELINE.json configuration the chosen shape corresponds to (label only)."""
    if (a.width, a.height) == (1280, 720) and T == 2:
        return "BASELINE.json configs[1]" if a.guidance > 1 else "BASELINE.json configs[2] (distilled: 1 forward/step)"
    if (a.width, a.height) == (1280, 720) and T == 8:
        return "BASELINE.json configs[3] shape (temporal reasoning, 8 latent frames)"
    if (a.width, a.height) == (1584, 1056):
        return ("BASELINE.json configs[4] (fp8 GEMM mode: fp8 weights / activations in the six large Linears, attention in bf16)"
                if a.fp8 else "BASELINE.json configs[4] shape, run in bf16")
    return "non-BASELINE shape"


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    from chronoedit_amd import ops
    from chronoedit_amd.pipeline import GraphedDenoiser, denoise_step, make_cfg_inputs
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.flops import dit_flops_per_forward

    ops.lib()  # fail loudly if the HIP library is missing
    if a.attn_kernel:
        ops.set_attention_waves(a.attn_kernel)
    model = build_model(a.layers, dev)
    model.cache_context = a.cache_context
    if a.fp8:
        model.enable_fp8_gemms()
    ulysses = world > 1 and a.parallel == "ulysses"
    if ulysses:
        model.enable_sequence_parallel()
        a.sequential_cfg = True  # the token shard is per sample
    T, h, w = a.frames, a.height // 8, a.width // 8
    N = T * (h // 2) * (w // 2)
    g = torch.Generator(device=dev).manual_seed(42 + (0 if ulysses else rank))  # Ulysses: replicated inputs
    latents = torch.randn((1, 16, T, h, w), generator=g, device=dev, dtype=torch.float32)
    condition = torch.randn((1, 20, T, h, w), generator=g, device=dev).to(torch.bfloat16)
    prompt = torch.randn((1, 512, 4096), generator=g, device=dev)
    prompt[:, 64:] = 0
    negative = torch.randn((1, 512, 4096), generator=g, device=dev)
    negative[:, 64:] = 0
    prompt, negative = prompt.to(torch.bfloat16), negative.to(torch.bfloat16)
    image = torch.randn((1, 257, 1280), generator=g, device=dev).to(torch.bfloat16)
    sched = FlowUniPCMultistepScheduler(flow_shift=5.0)
    total = a.warmup + a.steps + (0 if a.no_profile else 1)
    sched.set_timesteps(max(50, total), device=dev)
    fwd_per_step = 2 if a.guidance > 1.0 else 1

    cfg_inputs = make_cfg_inputs(prompt, negative, image)  # resident before the timed region, like every other input

    def one_step(i):
        denoise_step(model, sched, latents, condition, sched.timesteps[i], prompt, negative, image, a.guidance,
                     batch_cfg=not a.sequential_cfg, cfg_inputs=cfg_inputs)

    graphed = None
    if a.graph:
        sched._step_index = 0
        graphed = GraphedDenoiser(model, sched, latents, condition, prompt, negative, image, a.guidance, batch_cfg=not a.sequential_cfg)
        eager_step = one_step

        def one_step(i):  # noqa: F811
            graphed.step(i)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(a.warmup + i)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(latents).all().item())

    # ---- per-kernel HIP-event profile of ONE more step (outside the timed region) -> roofline of the dominant kernel
    roofline = None
    breakdown = None
    if not a.no_profile and ulysses and rank != 0:
        (eager_step if a.graph else one_step)(a.warmup + a.steps)  # the step has collectives: every rank must take part
    if not a.no_profile and rank == 0:
        with ops.profile() as prof:
            (eager_step if a.graph else one_step)(a.warmup + a.steps)
        summ = prof.summary()
        tot = sum(d["total_ms"] for d in summ.values())
        breakdown = {k: {"n": d["n"], "avg_ms": round(d["avg_ms"], 4), "share": round(d["total_ms"] / tot, 4),
                         "tflops": round(d["work"] / (d["avg_ms"] * 1e-3) / 1e12, 1) if k.startswith(("gemm", "attention")) else None,
                         "GBps": round(d["work"] / (d["avg_ms"] * 1e-3) / 1e9, 1) if k.startswith(("ln_", "rmsnorm")) else None}
                     for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        dom = max((k for k in summ if k.startswith(("gemm", "attention"))), key=lambda k: summ[k]["total_ms"])
        ach = summ[dom]["work"] / (summ[dom]["avg_ms"] * 1e-3) / 1e12
        traffic, traffic_src = _pmc_traffic(dom)
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launches": summ[dom]["n"], "avg_ms": round(summ[dom]["avg_ms"], 4),
                    # context, not the denominator: what a loop of nothing but MFMAs sustains on this chip when the operands
                    # toggle (power-limited clock; tools/probes/mfma_rate_probe.hip, profiles/r01_mfma_rate_probe.txt)
                    "sustained_mfma_only_random_operands": {"32x32x16": 2030.0, "16x16x32": 2160.0, "unit": "TFLOP/s",
                                                            "source": "profiles/r01_mfma_rate_probe.txt"}}

    # ---- secondary figure, outside the timed region: the same step with the step-invariant text / image K/V projections
    # computed once and reused (identical results, 1.4 % fewer flops) - reported beside `value`, never as it
    cached_rate = None
    if rank == 0 and not a.cache_context and not a.graph and world == 1:
        model.cache_context = True
        base = a.warmup + a.steps + (0 if a.no_profile else 1)
        one_step(base)  # fills the cache
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(2):
            one_step(base + 1 + i)
        torch.cuda.synchronize()
        cached_rate = round(2 / (time.perf_counter() - tc), 4)
        model.cache_context = False

    # ---- secondary figure, outside the timed region: the same step in the fp8 GEMM mode (BASELINE.json configs[4] arithmetic,
    # DESIGN.md section 9) - a different precision, reported beside the bf16 `value`, never as it
    fp8_rate = None
    if rank == 0 and not a.fp8 and not a.graph and world == 1 and not a.no_fp8_leg:
        model.enable_fp8_gemms()
        base = a.warmup + a.steps + 4
        one_step(base)  # packs the e4m3 weights
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(2):
            one_step(base + 1 + i)
        torch.cuda.synchronize()
        fp8_rate = round(2 / (time.perf_counter() - tc), 4)
        model.enable_fp8_gemms(False)

    # ---- VAE encode + decode at the same resolution (once per edit) -> composed sec/edit for the 50-step schedule
    vae_s = None
    if not a.no_vae and rank == 0 and world == 1:
        from chronoedit_amd.vae import AutoencoderKLWan
        vae = AutoencoderKLWan.random_init(dev, seed=4321)
        nf = 4 * (T - 1) + 1
        vid = (torch.rand(1, 3, nf, a.height, a.width, device=dev) * 2 - 1).to(torch.bfloat16)
        zl = torch.randn(1, 16, T, h, w, device=dev).to(torch.bfloat16)
        vae.encode(vid), vae.decode(zl)  # warm-up
        torch.cuda.synchronize()
        tv = time.perf_counter()
        vae.encode(vid).latent_dist.mode()
        torch.cuda.synchronize()
        te = time.perf_counter() - tv
        tv = time.perf_counter()
        vae.decode(zl, return_dict=False)
        torch.cuda.synchronize()
        vae_s = {"encode_s": round(te, 4), "decode_s": round(time.perf_counter() - tv, 4)}

    # ---- conditioning encoders (once per edit): UMT5-XXL on the positive + negative prompt padded to 512 tokens, CLIP ViT-H/14
    enc_s = None
    if not a.no_encoders and rank == 0 and world == 1:
        from chronoedit_amd.clip_vision import CLIPVisionModel
        from chronoedit_amd.umt5 import UMT5EncoderModel, t5_prompt_embeds
        torch.manual_seed(0)
        te_model = UMT5EncoderModel(device=dev)   # architecture defaults = google/umt5-xxl encoder; random-init weights
        ie_model = CLIPVisionModel(device=dev)    # ViT-H/14 defaults
        ids = torch.randint(2, 256384, (2, 512), device=dev)
        am = torch.zeros((2, 512), dtype=torch.long, device=dev)
        am[0, :64], am[1, :20] = 1, 1
        px = torch.randn(1, 3, 224, 224, device=dev)
        t5_prompt_embeds(te_model, ids, am), ie_model(pixel_values=px, output_hidden_states=True)  # warm-up (packs weights)
        torch.cuda.synchronize()
        tv = time.perf_counter()
        pe = t5_prompt_embeds(te_model, ids, am)
        torch.cuda.synchronize()
        tt_ = time.perf_counter() - tv
        tv = time.perf_counter()
        ie = ie_model(pixel_values=px, output_hidden_states=True).hidden_states[-2]
        torch.cuda.synchronize()
        enc_s = {"text_s": round(tt_, 4), "image_s": round(time.perf_counter() - tv, 4),
                 "finite": bool(torch.isfinite(pe.float()).all().item() and torch.isfinite(ie.float()).all().item())}
        del te_model, ie_model

    if rank == 0:
        steps_per_s = a.steps / dt * (1 if ulysses else world)
        fl = dit_flops_per_forward(N, num_layers=a.layers) * fwd_per_step
        out = {
            "metric": "denoising-steps/sec", "value": round(steps_per_s, 4), "unit": f"denoising-steps/sec (ChronoEdit-14B, {a.width}x{a.height})",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if ulysses else "weak", "vs_baseline": None,
            "dtype": "fp8 e4m3 GEMMs (fp32 accumulate), bf16 attention / norms / residual" if a.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"ChronoEdit-14B DiT ({a.layers} blocks), {a.width}x{a.height}, {T} latent frames (N={N} tokens), "
                                   f"guidance {a.guidance} ({fwd_per_step} forwards/step) + CFG + flow-UniPC update; "
                                   + _baseline_config_name(a, T),
                       "tokens": N, "forwards_per_step": fwd_per_step, "parallelism": f"ulysses sp{world}" if ulysses else f"replica x{world}",
                       "cfg": "sequential (2 x B=1)" if a.sequential_cfg else "batched (1 x B=2)",
                       "context_cache": bool(a.cache_context)},
            "model_tflops_per_step": round(fl / 1e12, 2),
            "achieved_tflops_per_gpu": round(fl * a.steps / dt / 1e12 / (world if ulysses else 1), 1),
            "mfma_roofline_frac_whole_step": round(fl * a.steps / dt / 1e12 / (world if ulysses else 1) / PEAK_BF16_TFLOPS, 4),
            "finite": finite,
            "launch": "hipGraph replay" if a.graph else "eager",
            "steps_per_sec_with_context_kv_cache": cached_rate,
            "steps_per_sec_fp8_gemm_mode": fp8_rate,
            "vae": vae_s,
            "encoders": enc_s,
            "sec_per_edit_50_steps": None if vae_s is None else round(
                50 * dt / a.steps + vae_s["encode_s"] + vae_s["decode_s"] + (0.0 if enc_s is None else enc_s["text_s"] + enc_s["image_s"]), 2),
            "roofline": roofline,
            "kernel_breakdown": breakdown,
        }
        if a.layers != 40:
            out["invalid"] = "reduced depth (debug run)"
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, fwd_per_step)
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
