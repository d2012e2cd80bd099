#!/usr/bin/env python
"""bench.py — denoising-steps/sec of the ChronoEdit-14B hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one iteration of ChronoEditPipeline.__call__'s loop
(/root/reference/chronoedit_diffusers/pipeline_chronoedit.py:695-756): guidance 5.0 -> TWO DiT forwards + CFG + flow-UniPC
update.  Synthetic (seeded) weights of the real 14B architecture and synthetic inputs — there are no checkpoints offline.
Inputs are resident in HBM before the timed region; nothing is cached across steps unless --cache-context is given (then the
step-invariant text/image K/V are reused, and the JSON says so).

N = 1: BASELINE.json configs[1] — bf16, 1280x720, 5 pixel frames -> latents [1,16,2,90,160] -> 7200 tokens.
N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): BASELINE.json configs[3] — the temporal-reasoning shape,
8 latent frames = 28 800 tokens, ONE edit with the token axis sharded over the N GPUs (Ulysses: three all-to-all per self-attention
over xGMI, chronoedit_amd/parallel.py), "scaling": "strong".  The line also carries, measured in the same run outside the timed
region: the same workload on ONE GPU (rank 0, so that the strong-scaling speed-up can be read off the line itself) and the
replica figure (N independent configs[1] edits, weak scaling).  `--parallel replica` makes the replica mode the headline instead.
How the N ranks are split (tools/scaling_model.py): N = 2 - the guidance pair split, one whole forward per GPU
(`--cfg-parallel`, the default there); N >= 4 - one N-way Ulysses group with the guidance pair batched INSIDE it (B = 2 per sharded
forward on the blocked-layout kernels); `--cfg-parallel` / `--no-cfg-parallel` / `--sequential-cfg` select the other splits.
Timing: barrier + synchronize on both sides, MAX over ranks.

The JSON line carries:
  roofline         the single largest kernel launch shape (the batched self-attention at 7200 tokens), algorithmic FLOPs / mean
                   launch duration measured with HIP events on the launch stream in one extra profiled step after the timed
                   region, vs 2.5 PFLOP/s dense bf16; `traffic` from the committed rocprofv3 --pmc passes under profiles/;
                   RULE: `roofline.kernel` = the launch LABEL (shape) with the largest total time in the profiled step - the batched self-attention;
                   the largest kernel SYMBOL (gemm_bf16_384<EPI_GATE_RES>, three shapes) is covered by roofline_family;
  roofline_family  the same accounting for ALL launches of the large-tile GEMM kernels together (72 % of a step);
  cpu_baseline     the reference's OWN ChronoEditTransformerBlock (transformer_chronoedit.py:215-295, executed from the build output
                   oracle/_ref/transformer_ref.bin: "kind": "reference"; the oracle port beside it, and alone - "kind": "port" - where that file
                   was not built) timed on this host's granted cores on ONE transformer block at the same token count (rank 0, at every N; one
                   warm-up + median of three, a single bounded run at 28 800 tokens), extrapolated to steps/sec; cpu_config0 = BASELINE configs[0];
  N > 1            `python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run; the headline is printed at once in a
                   "preliminary" line, the complete line (sharded sec/edit of the temporal-reasoning edits, cpu_baseline, rccl.model_prediction) last;
  sec_per_edit     MEASURED end to end through ChronoEditPipeline for configs[2] (8-step distilled schedule, guidance 1) and,
                   with --full-edit, configs[1] (50 steps); the composed 50-step figure is labelled as composed.
"""
import argparse
import json
import os
import statistics
import sys
import time

# dmabuf IPC is the only kind this host driver supports: without it RCCL's hipIpcGetMemHandle fails at N > 1 (already exported on the GPU
# boxes; a default here keeps a bare `torchrun bench.py` working - it must be set before the HIP runtime comes up)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0  # MI355X dense fp8 MFMA (MI355X_MICROARCH.md) - the denominator of every kernel computing in e4m3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=40, help="(debug only) fewer blocks => result marked invalid")
    ap.add_argument("--frames", type=int, default=None, help="latent frames: 2 (edit; default on one GPU / replicas) or 8 "
                                                             "(temporal reasoning; default for the Ulysses mode)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--cache-context", action="store_true", help="reuse step-invariant text/image K/V across steps")
    ap.add_argument("--sequential-cfg", action="store_true", help="two B=1 forwards per step instead of one batched B=2 forward")
    ap.add_argument("--parallel", choices=["auto", "replica", "ulysses"], default="auto",
                    help="N>1: 'ulysses' (default via auto) = ONE edit with the token axis sharded over the GPUs (all-to-all over "
                         "RCCL/xGMI, strong scaling); 'replica' = independent edits per GPU (weak scaling)")
    ap.add_argument("--cfg-parallel", dest="cfg_parallel", action="store_true", default=None,
                    help="sharded mode: the cond / uncond passes side by side on two (N/2)-way Ulysses groups (default on 2 GPUs)")
    ap.add_argument("--no-cfg-parallel", dest="cfg_parallel", action="store_false", help="sharded mode: one N-way Ulysses group (guidance pair batched inside it; --sequential-cfg: one pass after the other)")
    ap.add_argument("--no-transposed-v", action="store_true",
                    help="(A/B) fused q|k|v GEMM + register-staged attention kernel instead of V^T from the swapped GEMM + LDS-DMA staging")
    ap.add_argument("--graph", action="store_true", help="replay one hipGraph-captured step instead of launching eagerly")
    ap.add_argument("--owned-comm", action="store_true",
                    help="sharded mode: run the exchanges on the RCCL communicator owned by libchronoedit_hip (ce_comm_*; parallel.OwnedComm) instead "
                         "of torch.distributed's - the form under which --graph can capture the sharded step")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE encode/decode timing used for the sec/edit figure")
    ap.add_argument("--no-encoders", action="store_true", help="skip the UMT5 / CLIP timing used for the sec/edit figure")
    ap.add_argument("--no-edit", action="store_true", help="skip the measured 8-step end-to-end edit")
    ap.add_argument("--full-edit", dest="full_edit", action="store_true", default=True,
                    help="MEASURE the 50-step configs[1] edit end to end (~17 s; default on one GPU)")
    ap.add_argument("--no-full-edit", dest="full_edit", action="store_false", help="skip the measured 50-step edit")
    ap.add_argument("--no-fp8-leg", action="store_true", help="skip the secondary fp8-GEMM-mode timing")
    ap.add_argument("--no-fp8-config4", action="store_true", help="skip the BASELINE configs[4] leg (1584x1056, fp8, 3 steps) of the one-GPU line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-only", action="store_true",
                    help="no GPU: only the CPU legs (cpu_baseline at N = 7200 and BASELINE configs[0] at N = 512; with /root/reference present also "
                         "the reference's own transformer class) - what profiles/r03_cpu_legs.json holds")
    ap.add_argument("--reasoning-edit", dest="reasoning_edit", action="store_true", default=True,
                    help="MEASURE temporal-reasoning edits end to end on this GPU (29 pixel frames: 8 latent frames, truncated to 2 after "
                         "num_temporal_reasoning_steps steps; two decodes), one per entry of --reasoning-steps (default on one GPU: ~36 s + ~102 s)")
    ap.add_argument("--no-reasoning-edit", dest="reasoning_edit", action="store_false", help="skip the measured temporal-reasoning edits")
    ap.add_argument("--reasoning-steps", type=str, default="10,50",
                    help="num_temporal_reasoning_steps values of the measured reasoning edits, comma separated (50 = the reference's default: "
                         "never truncates, pipeline_chronoedit.py:700-709)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N>1: skip the single-GPU / replica legs beside the Ulysses line")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE.json configs[4] arithmetic: the six large Linears of every block in fp8 e4m3 (MX matrix instruction); "
                         "reported with dtype fp8, never the headline bf16 number")
    ap.add_argument("--fp8-gemms-only", action="store_true", help="with --fp8: keep the self-attention in bf16 (round-1 fp8 mode)")
    ap.add_argument("--fp8-policy", choices=["fast", "accurate"], default="fast",
                    help="with --fp8: which of the six large Linears of a block run in fp8 (ChronoEditTransformer3DModel.FP8_POLICIES): fast = all six, "
                         "accurate = all but the ungated cross-attention out-projection (4.0 x instead of 7.8 x the bf16 path's error per block)")
    ap.add_argument("--fp8-no-attn-quant-fusion", action="store_true",
                    help="with --fp8 (MX): A/B switch - the attention kernels write bf16 and a separate pass quantises the out-projections' operands")
    ap.add_argument("--fp8-row-scales", action="store_true",
                    help="with --fp8: the round-1..3 GEMM contract (one fp32 scale per token row / output channel) instead of OCP-MX block scales")
    ap.add_argument("--attn-kernel", type=int, default=0,
                    help="(tuning) self-attention kernel knob of ce_set_attention_waves: 0 auto, 8 plain, 64 sw-pipelined")
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


def _host_cores() -> int:
    """Cores this process may actually use: os.cpu_count() reports the HOST's (256 on the GPU boxes) even when the container's cgroup
    quota or affinity mask grants far fewer - 256 torch threads on a few granted cores made the round-3 CPU legs 20x slower than 8
    threads on 8 cores."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _median_time(fn, runs=3, warm=1, budget_s=None):
    """Median wall time of `runs` calls after `warm` untimed ones.  With `budget_s`: a bounded sample - when the warm-up call alone took more
    than a third of the budget it IS the sample (one cold run; the caller's text says so), and the timed runs stop once the budget is spent."""
    ts = []
    t_begin = time.perf_counter()
    for i in range(warm + runs):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if i >= warm:
            ts.append(dt)
        elif budget_s is not None and dt > budget_s / 3:
            return dt, [dt]
        if budget_s is not None and ts and time.perf_counter() - t_begin > budget_s:
            break
    return statistics.median(ts), ts


def _reference_transformer():
    """The reference's own transformer_chronoedit.py as built by oracle/build_ref.py into oracle/_ref/transformer_ref.bin (a marshalled
    code object: /root/reference does not exist on the GPU box, the build output does).  None when it was not built."""
    try:
        from oracle import build_ref
        return build_ref.load_transformer()
    except Exception:  # noqa: BLE001 - a baseline leg must never take the bench line down
        return None


def cpu_baseline(N: int, steps_fwd: int, budget_s: float = 45.0, grid=None):
    """The reference's OWN `ChronoEditTransformerBlock` (chronoedit_diffusers/transformer_chronoedit.py:215-295, executed from
    oracle/_ref/transformer_ref.bin over oracle/refshim's diffusers leaves: "kind": "reference") on this host's cores: one full-width block at
    N tokens, fp32 eager; one warm-up run, then the median of three; x40 blocks x forwards/step.  The CPU oracle
    ("port", oracle/dit_oracle.py) is timed on the same tensors right after it and reported beside it; it alone is the baseline when the
    reference build output is absent (kind "port")."""
    from oracle import dit_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    cfg = O.DiTConfig(num_layers=1)
    g = torch.Generator().manual_seed(1)
    p = {k: v for k, v in O.make_synthetic_params(cfg, seed=1).items() if k.startswith("blocks.0.")}
    x = torch.randn(1, N, cfg.inner_dim, generator=g)
    enc = torch.randn(1, 769, cfg.inner_dim, generator=g)
    temb6 = torch.randn(1, 6, cfg.inner_dim, generator=g) * 0.1
    T, hp, wp = grid if grid is not None else (2, 45, N // 90)  # (latent frames, patch rows, patch columns): 720p edit shape by default
    shaped = T * hp * wp == N
    rot = O.rope_table(cfg, T, 2 * hp, 2 * wp) if shaped else None
    t_start = time.perf_counter()
    ref = None
    mod = _reference_transformer()
    with torch.no_grad():
        if mod is not None and shaped:
            blk = mod.ChronoEditTransformerBlock(cfg.inner_dim, cfg.ffn_dim, cfg.num_attention_heads, cfg.qk_norm, cfg.cross_attn_norm, cfg.eps,
                                                 cfg.added_kv_proj_dim).eval()
            sd = {k[len("blocks.0."):]: v for k, v in p.items()}
            blk.load_state_dict(sd, strict=True, assign=True)
            rope = mod.ChronoEditRotaryPosEmbed(cfg.attention_head_dim, tuple(cfg.patch_size), cfg.rope_max_seq_len,
                                                temporal_skip_len=cfg.rope_temporal_skip_len)
            rot_ref = rope(torch.empty(1, 1, T, 2 * hp, 2 * wp))
            dt_r, times_r = _median_time(lambda: blk(x, enc, temb6, rot_ref), budget_s=budget_s)
            ref = (dt_r, times_r, None)
            del blk
        # the port beside it: the full median-of-three when it is the only baseline, one timed run after a warm-up when the reference
        # class was timed (and less if the host is slow: the whole leg stays inside `budget_s`)
        runs = 3 if ref is None else (1 if time.perf_counter() - t_start > budget_s / 2 else 2)
        if ref is not None and time.perf_counter() - t_start > budget_s:
            # one call of the reference class already spent the budget (N = 28 800 on the N > 1 lines): the port beside it is the N = 1 line's business
            dt_p, times_p = None, []
        else:
            dt_p, times_p = _median_time(lambda: O.block_forward(p, 0, cfg, x, enc, temb6, rot), runs=runs, budget_s=budget_s)
    port = None if dt_p is None else {
        "seconds_per_block": round(dt_p, 3), "runs": [round(t, 2) for t in times_p], "steps_per_sec": 1.0 / (dt_p * 40 * steps_fwd),
        "what": "oracle/dit_oracle.block_forward (the CPU restatement the parity tests check against), same tensors, same threads"}
    if ref is None:
        per_step = dt_p * 40 * steps_fwd
        return {"value": 1.0 / per_step, "unit": "denoising-steps/sec", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
                "sample": f"1 of 40 DiT blocks, N={N}, fp32 torch-CPU oracle, median of {len(times_p)} after 1 warm-up = {dt_p:.2f} s "
                          f"(runs {', '.join(f'{t:.2f}' for t in times_p)}); x40 blocks x{steps_fwd} forwards/step.  kind = port (oracle/dit_oracle.py): "
                          "oracle/_ref/transformer_ref.bin (the reference's own class, built by __graft_entry__.build() where /root/reference exists) "
                          "was not found on this box", "port": port}
    dt_r, times_r, _ = ref
    per_step = dt_r * 40 * steps_fwd
    return {"value": 1.0 / per_step, "unit": "denoising-steps/sec", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "reference",
            "sample": f"1 of 40 DiT blocks, N={N}, fp32 eager: the reference's own ChronoEditTransformerBlock.forward (chronoedit_diffusers/"
                      f"transformer_chronoedit.py:215-295 compiled into oracle/_ref/transformer_ref.bin; diffusers leaf modules from oracle/refshim) "
                      f"with its own ChronoEditRotaryPosEmbed table, torch-CPU on {cores} threads, " +
                      (f"median of {len(times_r)} after 1 warm-up" if len(times_r) > 1 else f"ONE run (bounded sample: a single call already exceeds a third of the {budget_s:.0f} s budget)") +
                      f" = {dt_r:.2f} s (runs {', '.join(f'{t:.2f}' for t in times_r)}); x40 blocks x{steps_fwd} forwards/step",
            "port": port, "reference_over_port_time": None if dt_p is None else round(dt_r / dt_p, 3)}


def cpu_config0(with_reference: bool = True, depths=(1, 2, 4)):
    """BASELINE.json configs[0] on the host cores (SURVEY section 8d "Config 1"): the 14B width at 256x256 / 5 pixel frames
    (latents [1,16,2,32,32], N = 512 tokens), fp32 eager, 4 steps x 2 forwards.  Full depth in fp32 is 61 GiB of weights, so
    the whole forward (embedders, L blocks, head) is timed at L = 1, 2 (inside the default bench run) or 1, 2, 4 (`--cpu-only`) and the
    per-block time is the slope; 40 blocks are composed from it.  "port" = oracle/dit_oracle.py; "reference" = the reference's own ChronoEditTransformer3DModel
    (chronoedit_diffusers/transformer_chronoedit.py) with oracle/refshim standing in for the diffusers leaf modules - only where
    /root/reference exists (the build container; not the GPU box)."""
    from oracle import dit_oracle as O
    cores = _host_cores()
    torch.set_num_threads(cores)
    cfg4 = O.DiTConfig(num_layers=max(depths))
    p4 = O.make_synthetic_params(cfg4, seed=1234)
    lat, text, image = O.make_synthetic_inputs(cfg4, 2, 32, 32, dtype=torch.float32)
    ts = torch.tensor([637])
    out = {"workload": "ChronoEdit-14B width, 256x256 px, 5 pixel frames = 2 latent frames (N = 512), fp32, CPU; 4 steps x 2 forwards per edit",
           "cores": cores}

    def sub(L):
        keep = lambda k: (not k.startswith("blocks.")) or int(k.split(".")[1]) < L
        return {k: v for k, v in p4.items() if keep(k)}

    def fit(Ls, times):  # forward(L) = a + b L  (least squares over the depths timed)
        n = len(Ls)
        mL, mt = sum(Ls) / n, sum(times) / n
        b = sum((l - mL) * (t - mt) for l, t in zip(Ls, times)) / sum((l - mL) ** 2 for l in Ls)
        return mt - b * mL, b

    legs = {}
    t_start = time.perf_counter()
    budget_s = 120.0  # the whole bench must stay within minutes on any host: depths beyond the budget are dropped (>= 2 are needed for the slope)
    with torch.no_grad():
        t_port, Ls_port = [], []
        for L in depths:
            if Ls_port and len(Ls_port) >= 2 and time.perf_counter() - t_start > budget_s:
                break
            cfg = O.DiTConfig(num_layers=L)
            pl = sub(L)
            dt, _ = _median_time(lambda: O.dit_forward(pl, cfg, lat, ts, text, image), runs=2, warm=0 if Ls_port else 1)
            t_port.append(dt)
            Ls_port.append(L)
        legs["port"] = (Ls_port, t_port)
        mod = _reference_transformer() if with_reference else None
        if mod is not None:
            from oracle import gen_golden as G  # (build_reference_model only: constructs the reference class and loads the synthetic state dict)
            t_ref = []
            for L in Ls_port:
                if t_ref and time.perf_counter() - t_start > 2 * budget_s:
                    break
                m = G.build_reference_model(mod, O.DiTConfig(num_layers=L), sub(L))
                dt, _ = _median_time(lambda: m(lat, ts, text, image, return_dict=False), runs=2, warm=0 if t_ref else 1)
                t_ref.append(dt)
                del m
            if len(t_ref) >= 2:
                legs["reference"] = (Ls_port[:len(t_ref)], t_ref)
    for kind, (Ls, tl) in legs.items():
        a0, b = fit([float(l) for l in Ls], tl)
        fwd40 = a0 + 40 * b
        out[kind] = {"depths": list(Ls), "forward_s_at_depths": [round(t, 3) for t in tl], "per_block_s": round(b, 4), "outside_blocks_s": round(max(a0, 0.0), 4),
                     "forward_40_blocks_s": round(fwd40, 2), "denoising_steps_per_sec": round(1.0 / (2 * fwd40), 5),
                     "sec_per_4_step_edit_dit_only": round(8 * fwd40, 1), "kind": kind}
    return out


def _power_sample(step_fn, first_index: int, n_steps: int = 6):
    """Package power and shader clock WHILE the step runs (best effort, one GPU, outside the timed region): enqueue `n_steps` more steps, ask rocm-smi
    once half a second in, then synchronise.  profiles/r05_power_clock_trace.txt is the same reading sampled every 0.5 s: both the bf16 and the fp8 step
    sit at ~1.35 kW of the 1.4 kW cap with the shader clock pulled to 1.9 / 2.2 GHz of 2.4 - the step time is energy per step over the cap.
    None when rocm-smi is missing or says nothing parsable."""
    import re
    import shutil
    import subprocess
    smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if smi is None:
        return None
    try:
        for i in range(n_steps):
            step_fn(first_index + i)
        time.sleep(0.5)
        txt = subprocess.run([smi, "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
        torch.cuda.synchronize()
        pw = re.search(r"GPU\[0\].*?Current Socket Graphics Package Power \(W\): ([0-9.]+)", txt)
        cap = re.search(r"GPU\[0\].*?Max Graphics Package Power \(W\): ([0-9.]+)", txt)
        ck = re.search(r"GPU\[0\].*?sclk clock level: \w+: \((\d+)Mhz\)", txt)
        if not pw or not ck:
            return None
        return {"package_power_w": float(pw.group(1)), "power_cap_w": float(cap.group(1)) if cap else None, "sclk_mhz": int(ck.group(1)), "sclk_max_mhz": 2400,
                "how": "one rocm-smi reading 0.5 s into six more steps after the timed region (device 0)",
                "see": "profiles/r05_power_clock_trace.txt: the step runs at the package's power cap; its time is energy per step over the cap"}
    except Exception:  # noqa: BLE001 - a diagnostic must never take the line down
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            pass
        return None


def _pmc_traffic(kernel_label: str):
    """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE
    are collected in their own runs, tools/gpu_pmc.sh; they cannot be read live from inside this process).  None when the
    shape of this run has no committed measurement."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                rec = json.load(f).get(kernel_label)
        except (OSError, ValueError):
            rec = None
        if rec:
            return rec["fetch_bytes"] + rec["write_bytes"], f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch)"
    return None, None


def _kernel_symbol(label: str, work: float):
    """The kernel symbol a profiled launch label runs on (ops.py labels are shape + epilogue; the dispatcher's choice is replayed through the
    library's own pure function ce_gemm_bf16_tile_rows).  None for the row passes."""
    import re
    from chronoedit_amd import ops
    m = re.match(r"^gemm_(\d+)x(\d+)x(\d+)_epi(\d+)$", label)
    if m:
        M, N, K, e = (int(x) for x in m.groups())
        if e in (4, 5) or M * N < 256 * 256 * 128 or K % 128:
            return f"gemm_bf16_128<{e}>"
        rows = ops.lib().ce_gemm_bf16_tile_rows(M, N, K, 256, ops.GEMM_WS_BYTES)
        # (384 rows: gemm_bf16_384<EPI, 12>, written <EPI> as in rounds 4-6's profiles; 288 rows: the NF = 9 instantiation of the same kernel)
        return f"gemm_bf16_384<{e}>" if rows == 384 else f"gemm_bf16_384<{e}, NF=9 (288 rows)>" if rows == 288 else f"gemm_bf16_w4<{e}>"
    m = re.match(r"^gemm_mxfp8_\d+x\d+x\d+_(epi(\d+)|gelu_quant)$", label)
    if m:
        return f"gemm_fp8_w4<{m.group(2) if m.group(2) is not None else 7}, MX>"
    if label.startswith("gemm_fp8_"):
        return "gemm_fp8_w4<per-row scales>"
    if label.startswith("attention_mxfp8"):
        return "attn_fwd_mxfp8_sp_kernel"
    m = re.match(r"^attention_(\d+)x(\d+)\+(\d+)", label)
    if m:
        return "attn_fwd_sp_kernel<cross: two key segments>" if int(m.group(3)) > 0 else "attn_fwd_sp_kernel<self, V^T>"
    if label.startswith(("gemm", "attention")):
        return label.split("_")[0] + "_other"
    return None


def _baseline_config_name(a, T) -> str:
    """Which BASELINE.json configuration the chosen shape corresponds to (label only)."""
    if (a.width, a.height) == (1280, 720) and T == 2:
        return "BASELINE.json configs[1]" if a.guidance > 1 else "BASELINE.json configs[2] (distilled: 1 forward/step)"
    if (a.width, a.height) == (1280, 720) and T == 8:
        return "BASELINE.json configs[3] (temporal reasoning, 8 latent frames)"
    if (a.width, a.height) == (1584, 1056):
        return ("BASELINE.json configs[4] (fp8 weights / activations in the six large Linears" +
                (", attention in bf16)" if a.fp8_gemms_only else " + MXFP8 self-attention)") if a.fp8 else "BASELINE.json configs[4] shape, run in bf16")
    return "non-BASELINE shape"


class Workload:
    """Resident inputs of one edit at a given latent shape (seeded)."""

    def __init__(self, dev, T, h, w, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.T, self.h, self.w = T, h, w
        self.N = T * (h // 2) * (w // 2)
        self.latents = torch.randn((1, 16, T, h, w), generator=g, device=dev, dtype=torch.float32)
        self.condition = torch.randn((1, 20, T, h, w), generator=g, device=dev).to(torch.bfloat16)
        prompt = torch.randn((1, 512, 4096), generator=g, device=dev)
        prompt[:, 64:] = 0
        negative = torch.randn((1, 512, 4096), generator=g, device=dev)
        negative[:, 64:] = 0
        self.prompt, self.negative = prompt.to(torch.bfloat16), negative.to(torch.bfloat16)
        self.image = torch.randn((1, 257, 1280), generator=g, device=dev).to(torch.bfloat16)


def _self_launch(n: int):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): re-execute this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` - the form the driver
    uses for N > 1 - so the same verb works at every N.  Replaces the process (exit code and stdout are the launcher's).  A box with fewer
    than N GPUs gets ONE parsable error line and exit code 2 instead of N ranks dying in set_device."""
    if os.environ.get("CE_BENCH_SELF_LAUNCHED"):
        raise RuntimeError("bench.py re-launched itself but still sees no WORLD_SIZE: torch.distributed.run did not set the rank environment")
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("CE_BENCH_TEST_BACKEND"):
        print(json.dumps({"metric": "denoising-steps/sec", "value": None, "n_gpus": n,
                          "error": f"--gpus {n} but this box exposes {have} GPU(s)"}), flush=True)
        sys.exit(2)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, CE_BENCH_SELF_LAUNCHED="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def main():
    a = parse()
    if a.cpu_only:
        out = {"cpu_baseline": cpu_baseline(7200, 2), "cpu_config0": cpu_config0()}
        print(json.dumps(out), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = ctl = None
    # CE_BENCH_TEST_BACKEND=gloo (tools/gpu_r2_g.sh only): the N > 1 code path of this file on a ONE-GPU box - all ranks share GPU 0
    # and the collectives are host-staged.  The line it prints is marked and is not a measurement of anything.
    test_backend = os.environ.get("CE_BENCH_TEST_BACKEND")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_backend:
            local = 0
        torch.cuda.set_device(local)
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("CE_BENCH_PG_TIMEOUT_S", "600")))
        dist.init_process_group(test_backend or "nccl", timeout=pg_timeout)  # "nccl" == RCCL on ROCm
        if not test_backend and (dist.get_world_size() != world or dist.get_backend() != "nccl"):
            raise RuntimeError(f"RCCL group came up with {dist.get_world_size()} ranks on backend {dist.get_backend()}, expected {world} on nccl")
        # control plane on its own host-side (gloo) group: the barriers, the MAX over ranks of the host clocks and the "did every rank get
        # through the sharded leg" vote never ride on RCCL, so a data-path failure still ends in a JSON line (and in the replica fallback)
        # (its timeout outlasts the data group's: a rank that failed early waits in the vote while its peers run into theirs)
        ctl = dist.new_group(backend="gloo", timeout=3 * pg_timeout + datetime.timedelta(minutes=5))
        if a.gpus != world:
            raise RuntimeError(f"--gpus {a.gpus} but the launcher started {world} ranks")
    elif a.gpus != 1:
        _self_launch(a.gpus)  # bare `python bench.py --gpus N`: becomes the torch.distributed.run launch the driver uses (never returns)
    else:
        torch.cuda.set_device(0)
    # CE_BENCH_ONE_RANK_SP=1 (tests/test_bench_multirank_gpu.py only): the SHARDED code path of this file - Ulysses exchanges on real RCCL, the
    # library-owned communicator (--owned-comm), the captured sharded step (--graph) - with a group of ONE rank on a one-GPU box.  Every
    # exchange runs as a real RCCL call (parallel.Ulysses(force=True)); the line is marked and measures nothing but that the path works.
    one_rank_sp = world == 1 and a.gpus == 1 and os.environ.get("CE_BENCH_ONE_RANK_SP") == "1"
    if one_rank_sp:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("CE_BENCH_PG_TIMEOUT_S", "600")))
        dist.init_process_group("nccl", rank=0, world_size=1, timeout=pg_timeout)
        ctl = dist.new_group(backend="gloo", timeout=pg_timeout)
    dev = torch.device("cuda", local if world > 1 else 0)

    from chronoedit_amd import ops
    from chronoedit_amd.flops import dit_flops_per_forward
    from chronoedit_amd.pipeline import GraphedDenoiser, denoise_step, make_cfg_inputs
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler

    ops.lib()  # fail loudly if the HIP library is missing
    if a.attn_kernel:
        ops.set_attention_waves(a.attn_kernel)
    model = build_model(a.layers, dev)
    model.cache_context = a.cache_context
    cache_flag = a.cache_context  # (the pipeline object of the sec/edit legs switches the per-edit context cache on for ITS edits)
    if a.no_transposed_v:
        model.enable_transposed_v(False)
    if a.fp8:
        model.fp8_fuse_attn_quant = not a.fp8_no_attn_quant_fusion
        model.enable_fp8_gemms(mx=not a.fp8_row_scales, policy=a.fp8_policy)
        if not a.fp8_gemms_only:
            model.enable_fp8_attention()
    mode = a.parallel
    if mode == "auto":
        mode = "ulysses" if (world > 1 or one_rank_sp) else "replica"
    ulysses = (world > 1 or one_rank_sp) and mode == "ulysses"
    # Default split of the N ranks (tools/scaling_model.py): on TWO GPUs the guidance pair is split - each
    # GPU runs one of the two forwards whole, no all-to-all at all, one 3.7 MB exchange per step - because a 2-rank Ulysses group
    # talks over ONE of the seven xGMI links (predicted 0.93 vs 0.69 steps/s); from four GPUs on ONE Ulysses group over all ranks, the
    # guidance pair batched inside it (blocked-layout kernels), has the links (3 resp. 7 per GPU) and wins.  --cfg-parallel / --no-cfg-parallel override.
    if a.cfg_parallel is None:
        a.cfg_parallel = bool(ulysses and world == 2 and a.guidance > 1)
    if a.cfg_parallel and not (ulysses and world % 2 == 0 and a.guidance > 1):
        raise RuntimeError("--cfg-parallel needs the Ulysses mode on an even number of GPUs with guidance > 1")
    T = a.frames if a.frames is not None else (8 if ulysses else 2)
    h, w = a.height // 8, a.width // 8
    if ulysses:
        if a.cfg_parallel:
            model.enable_cfg_parallel()
        else:
            model.enable_sequence_parallel(force=one_rank_sp, owned_comm=a.owned_comm)
    wl = Workload(dev, T, h, w, 42 + (0 if ulysses else rank))  # Ulysses: replicated inputs
    N = wl.N
    fwd_per_step = 2 if a.guidance > 1.0 else 1

    def make_stepper(wl_, sched_, graph=False, sequential=False):
        cfg_inputs = make_cfg_inputs(wl_.prompt, wl_.negative, wl_.image)  # resident before the timed region, like every input
        if graph:
            sched_._step_index = 0
            gd = GraphedDenoiser(model, sched_, wl_.latents, wl_.condition, wl_.prompt, wl_.negative, wl_.image, a.guidance,
                                 batch_cfg=not sequential, keep_warmup_step=False)  # every timed step() is a replay
            return lambda i: gd.step(i)
        return lambda i: denoise_step(model, sched_, wl_.latents, wl_.condition, sched_.timesteps[i], wl_.prompt, wl_.negative,
                                      wl_.image, a.guidance, batch_cfg=not sequential, cfg_inputs=cfg_inputs)

    def new_sched(n):
        s = FlowUniPCMultistepScheduler(flow_shift=5.0)
        s.set_timesteps(max(50, n), device=dev)
        return s

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(group=ctl)
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=ctl)
        return float(tt.item())

    def timed(step_fn, warm, steps, first=0):
        for i in range(warm):
            step_fn(first + i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(first + warm + i)
        sync_all()
        return max_over_ranks(time.perf_counter() - t0)

    total = a.warmup + a.steps + (0 if a.no_profile else 2)

    def all_ranks_ok(ok: bool) -> bool:
        """Did EVERY rank get here without an exception?  (host-side vote on the control group)"""
        if world == 1:
            return ok
        tt = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=ctl)
        return float(tt.item()) == 0.0

    def fall_back_to_replica(err: str):
        """The sharded leg threw (on this or another rank): say so in ONE JSON line, then run and print the `--parallel replica` line -
        the driver always gets a parsable last line, and a reader sees why it is not the sharded one."""
        nonlocal ulysses, mode, wl, N, T
        if rank == 0:
            print(json.dumps({"metric": "denoising-steps/sec", "value": None, "n_gpus": world, "error": err,
                              "config": {"workload": f"ONE configs[3] edit sharded over {world} GPUs (Ulysses over RCCL)"},
                              "note": "the sharded leg failed; the next line is the --parallel replica measurement of the same run"}), flush=True)
        model._sp = model._cfgp = None
        model.invalidate()
        ulysses, mode = False, "replica"
        T = a.frames if a.frames is not None else 2
        wl = Workload(dev, T, h, w, 42 + rank)
        N = wl.N

    sharded_error = None
    sched = new_sched(total)
    dt = finite = host_enqueue_ms = rccl = exchange_us = verify = None
    one_step = eager_step = None

    def headline_leg():
        nonlocal dt, finite, host_enqueue_ms, rccl, one_step, eager_step, exchange_us, verify
        one_step = make_stepper(wl, sched, graph=a.graph, sequential=a.sequential_cfg)
        eager_step = one_step if not a.graph else make_stepper(wl, sched, graph=False, sequential=a.sequential_cfg)
        if ulysses:
            model._sp.stats.update(all_to_all_calls=0, all_to_all_bytes_sent_off_rank=0)
            if os.environ.get("CE_BENCH_INJECT_SHARDED_FAILURE") == str(rank) or os.environ.get("CE_BENCH_INJECT_SHARDED_FAILURE") == "all":
                raise RuntimeError("injected failure of the sharded leg (CE_BENCH_INJECT_SHARDED_FAILURE; tests/test_bench_multirank_gpu.py)")
        dt = timed(one_step, a.warmup, a.steps)
        finite = bool(torch.isfinite(wl.latents).all().item())
        # how long the host needs to ISSUE one eager step (no wait): the margin by which a launch-per-kernel loop stays GPU-bound
        # (why the sharded loop, which cannot be hipGraph-captured on this stack, loses nothing by running eagerly)
        if not a.no_profile:
            sync_all()
            t0h = time.perf_counter()
            eager_step(a.warmup + a.steps)
            host_enqueue_ms = round((time.perf_counter() - t0h) * 1e3, 2)
            sync_all()
        if ulysses:
            st = model._sp.stats
            pair_batched = not a.cfg_parallel and not a.sequential_cfg and fwd_per_step == 2  # the guidance pair as one B = 2 sharded forward
            n_fwd = (a.warmup + a.steps) * (1 if (a.cfg_parallel or pair_batched) else fwd_per_step)
            rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "ulysses_group": model._sp.world,
                    "communicator": "library-owned (ce_comm_*: grouped ncclSend / ncclRecv on the step's streams)" if getattr(model._sp, "comm", None) is not None
                                    else "torch.distributed process group",
                    "cfg_parallel_groups": 2 if a.cfg_parallel else 1,
                    "all_to_all_per_layer_per_forward": st["all_to_all_calls"] / max(1, n_fwd * a.layers),
                    "bytes_sent_off_rank_per_layer_per_forward": st["all_to_all_bytes_sent_off_rank"] // max(1, n_fwd * a.layers),
                    "exchange": ("k|v all-to-all overlapped with the q projection; q; attention output (K-segmented operand of the out-projection)"
                                 if model._sp.world > 1 else "none inside a forward: each GPU runs one guidance pass whole") +
                                ("; one all_gather of the two predictions per step" if a.cfg_parallel else "")}
            exchange_us = time_exchanges()
            rccl["exchange_us_per_layer"] = exchange_us
            # what the scaling model (tools/scaling_model.py) predicts for every split of these ranks - the pair batched in
            # one group, two groups side by side, eager or as a captured loop (where the k|v exchange no longer hides behind the q projection) -
            # next to what this run chose, so the first real multi-GPU line can be read against it.  Full-size configs[3] only.
            if N == 28800 and a.layers == 40 and not test_backend:
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import scaling_model
                    rccl["model_prediction"] = scaling_model.predict(world)
                    rccl["model_prediction"]["this_run"] = (("cfg-parallel 2 x %d" % model._sp.world) if a.cfg_parallel else
                                                            ("ulysses %d" % world + (", B=2" if pair_batched else ""))) + (", hipGraph" if a.graph else ", eager")
                except Exception as e:  # noqa: BLE001 - a prediction must never take the line down
                    rccl["model_prediction"] = {"error": repr(e)}
            verify = sharded_result()

    def time_exchanges():
        """The three all-to-all exchanges of ONE layer (k|v, q, attention output), each alone on the engine's own send / receive buffers
        of this run's shape: median of 5 after one warm call, MAX over ranks; event-timed on the launch stream (RCCL) or, host-staged
        (test backend), by the host clock.  Outside the timed region."""
        sp = model._sp
        if sp is None or sp.world == 1:
            return None
        ws = list(model.engine()._ws.values())[-1]
        W = sp.world
        legs = {"k|v": (ws.send_kv, ws.recv_kv), "q": (ws.send_q, ws.recv_q), "output": (ws.att_g.view(W, -1, ws.att_g.shape[1]), ws.att_seg)}
        res = {}
        for name, (snd, rcv) in legs.items():
            ts = []
            for i in range(6):
                sync_all()
                if sp._host_staged:
                    t0 = time.perf_counter()
                    sp.all_to_all(snd, rcv)
                    torch.cuda.synchronize()
                    us = (time.perf_counter() - t0) * 1e6
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    sp.all_to_all(snd, rcv)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3
                if i:
                    ts.append(us)
            us = max_over_ranks(statistics.median(ts))
            off = snd.numel() * snd.element_size() * (W - 1) // W
            res[name] = {"us": round(us, 1), "bytes_sent_off_rank": off, "GBps_per_rank": round(off / us / 1e3, 1)}
        return res

    lat0_verify = None

    def sharded_result():
        """ONE sharded step from freshly seeded (replicated) inputs; the latents it leaves are compared with the unsharded step on the same
        inputs further down (rank 0), and - here - between the ranks: every rank must hold the same replicated result."""
        nonlocal lat0_verify
        wv = Workload(dev, T, h, w, 4242)
        lat0_verify = (wv, wv.latents.clone())
        sv = new_sched(3)
        make_stepper(wv, sv, sequential=a.sequential_cfg)(0)
        torch.cuda.synchronize()
        chk = float(wv.latents.double().abs().sum().item())
        lo, hi = -max_over_ranks(-chk), max_over_ranks(chk)
        lat0_verify = lat0_verify + (wv.latents.clone(),)
        return {"latents_abs_sum_spread_over_ranks": (hi - lo) / max(hi, 1e-30)}

    if ulysses:
        try:
            headline_leg()
            ok = True
        except Exception as e:  # noqa: BLE001 - anything: the line must still be printed
            import traceback
            sharded_error = f"rank {rank}: {type(e).__name__}: {e} | " + traceback.format_exc(limit=4).replace("\n", " / ")
            ok = False
            if rank != 0:  # said at once, from the rank it happened on: should the peers never reach the vote (a watchdog abort under RCCL) this line still exists
                print(json.dumps({"metric": "denoising-steps/sec", "value": None, "n_gpus": world, "rank": rank, "error": sharded_error}), flush=True)
        try:
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            sharded_error = (sharded_error or "") + f" | synchronize: {e!r}"
            ok = False
        if not all_ranks_ok(ok):
            fall_back_to_replica(sharded_error or "another rank failed in the sharded leg (its message is on its stderr)")
            sharded_error = sharded_error or "a peer rank failed"
            sched = new_sched(total)
            headline_leg()
    else:
        headline_leg()

    power = None
    if world == 1 and not one_rank_sp and not a.no_profile and not a.graph and total >= a.warmup + a.steps + 2:
        # (its own seeded inputs and scheduler: the timed workload's latents and schedule are left as the timed region left them)
        power = _power_sample(make_stepper(Workload(dev, T, h, w, 4343), new_sched(8), sequential=a.sequential_cfg), 0)
    # ---- per-kernel HIP-event profile of ONE more step (outside the timed region) -> roofline of the dominant kernel
    roofline = roofline_family = breakdown = None
    if not a.no_profile and ulysses and rank != 0:
        eager_step(a.warmup + a.steps + 1)  # the step has collectives: every rank must take part
    if not a.no_profile and rank == 0:
        with ops.profile() as prof:
            eager_step(a.warmup + a.steps + 1)
        summ = prof.summary()
        tot = sum(d["total_ms"] for d in summ.values())
        breakdown = {k: {"n": d["n"], "avg_ms": round(d["avg_ms"], 4), "share": round(d["total_ms"] / tot, 4),
                         "tflops": round(d["work"] / (d["avg_ms"] * 1e-3) / 1e12, 1) if k.startswith(("gemm", "attention")) else None,
                         "GBps": round(d["work"] / (d["avg_ms"] * 1e-3) / 1e9, 1) if k.startswith(("ln_", "rmsnorm", "rope_")) else None}
                     for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        # roofline.kernel = the largest SYMBOL of the step (the largest label - one shape + epilogue - was the attention at 20.9 % while
        # gemm_bf16_384<2> ran 28.4 % of the step under three labels): every matrix launch is mapped to the kernel symbol the dispatcher sends its
        # shape to, launches of one symbol are summed over shapes (achieved = sum of flops / sum of HIP-event time), the largest group is reported;
        # the largest single label stays in the line as `roofline_largest_label`
        groups = {}
        for k, d in summ.items():
            sym = _kernel_symbol(k, d["work"])
            if sym is None:
                continue
            g_ = groups.setdefault(sym, {"ms": 0.0, "fl": 0.0, "n": 0, "labels": []})
            g_["ms"] += d["total_ms"]
            g_["fl"] += d["work"] * d["n"]
            g_["n"] += d["n"]
            g_["labels"].append((d["total_ms"], k))
        dom_sym = max(groups, key=lambda k_: groups[k_]["ms"])
        G = groups[dom_sym]
        dom = max(G["labels"])[1]  # its largest shape
        ach = G["fl"] / (G["ms"] * 1e-3) / 1e12
        traffic = traffic_src = None
        for _, lb in sorted(G["labels"], reverse=True):  # the committed PMC traffic figure of the symbol's largest shape that has one (`traffic_of`)
            traffic, traffic_src = _pmc_traffic(lb)
            if traffic is not None:
                dom = lb
                break
        peak = PEAK_FP8_TFLOPS if ("fp8" in dom) else PEAK_BF16_TFLOPS  # attention_mxfp8_* / gemm_fp8_* run the fp8 MFMA
        roofline = {"kernel": dom_sym, "rule": "largest kernel SYMBOL of the profiled step by summed HIP-event time (launches of one symbol grouped over shapes / call sites)",
                    "labels": [k_ for _, k_ in sorted(G["labels"], reverse=True)], "share_of_step": round(G["ms"] / tot, 4),
                    "bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_of": dom if traffic is not None else None, "traffic_source": traffic_src,
                    "launches": G["n"], "avg_ms": round(G["ms"] / G["n"], 4),
                    # context, not the denominator: what a loop of nothing but MFMAs sustains on this chip when the operands
                    # toggle (power-limited clock; tools/probes/mfma_rate_probe.hip, profiles/r01_mfma_rate_probe.txt)
                    "sustained_mfma_only_random_operands": {"32x32x16": 2030.0, "16x16x32": 2160.0, "unit": "TFLOP/s",
                                                            "source": "profiles/r01_mfma_rate_probe.txt"}}
        lab = max((k for k in summ if k.startswith(("gemm", "attention"))), key=lambda k: summ[k]["total_ms"])
        roofline["largest_label"] = {"label": lab, "share_of_step": round(summ[lab]["total_ms"] / tot, 4), "avg_ms": round(summ[lab]["avg_ms"], 4),
                                     "achieved": round(summ[lab]["work"] / (summ[lab]["avg_ms"] * 1e-3) / 1e12, 1),
                                     "frac": round(summ[lab]["work"] / (summ[lab]["avg_ms"] * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if "fp8" in lab else PEAK_BF16_TFLOPS), 4)}
        roofline["by_symbol"] = {k_: {"share_of_step": round(v_["ms"] / tot, 4), "launches": v_["n"], "achieved": round(v_["fl"] / (v_["ms"] * 1e-3) / 1e12, 1),
                                      "frac": round(v_["fl"] / (v_["ms"] * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if "fp8" in k_ else PEAK_BF16_TFLOPS), 4)}
                                 for k_, v_ in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])[:8]}
        big = {k: d for k, d in summ.items() if k.startswith("gemm_") and "fp8" not in k and d["work"] >= 2.0 * 256 * 256 * 128 * 64}  # the 256-tile kernel's launches
        if big:
            fam_ms = sum(d["total_ms"] for d in big.values())
            fam_fl = sum(d["work"] * d["n"] for d in big.values())
            fam = fam_fl / (fam_ms * 1e-3) / 1e12
            roofline_family = {"kernel": "gemm_bf16_w4 + gemm_bf16_384 (every launch of the large-tile LDS-DMA GEMM in the step - one wave per SIMD, 256x256, 288x256 or 384x256 macro tile chosen per shape, all epilogues, split-K reduces included)", "bound": "mfma",
                               "achieved": round(fam, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fam / PEAK_BF16_TFLOPS, 4),
                               "launches": sum(d["n"] for d in big.values()), "total_ms": round(fam_ms, 3), "share_of_step": round(fam_ms / tot, 4)}

    single = dict(cached_rate=None, fp8_rate=None, fp8_config4=None, fp8_policies=None, vae_s=None, enc_s=None, edit8=None, edit50=None, edit_reasoning=None)
    if world == 1 and rank == 0 and not one_rank_sp:
        _single_gpu_secondaries(a, model, wl, new_sched, make_stepper, dev, T, h, w, single)

    # ---- N > 1: the headline is measured; say it NOW in a line marked preliminary (the legs below run whole edits and CPU baselines for minutes -
    # should one of them take the process down, the timed figure of this run is already on stdout), then measure sec/edit with the DiT sharded
    if world > 1 and rank == 0 and dt is not None:
        print(json.dumps({"metric": "denoising-steps/sec", "value": round(a.steps / dt * (1 if ulysses else world), 4), "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "scaling": "strong" if ulysses else "weak", "preliminary": True,
                          "note": "headline of this run, printed before the secondary legs; the COMPLETE line (same value + roofline, cpu_baseline, sec/edit) is the last line"}),
              flush=True)
    sharded_edit = None
    if ulysses and a.reasoning_edit and not a.no_edit and not a.no_vae and T == 8:
        try:
            sharded_edit = _sharded_edit_leg(a, model, dev, world, max_over_ranks, test_backend)
        except Exception as e:  # noqa: BLE001 - reported, never fatal to the line
            sharded_edit = {"error": f"rank {rank}: {type(e).__name__}: {e}"}

    # ---- N > 1, Ulysses headline: the same workload on ONE GPU (rank 0) and the replica (weak-scaling) figure, both outside
    # the timed region, so the line is self-contained
    single_same = replica = None
    if ulysses and not a.no_secondary:
        sp, cfgp = model._sp, model._cfgp
        model._sp = model._cfgp = None
        model.engine()._ws = {}
        if rank == 0:  # the denominator of the strong-scaling speed-up: median of three timed steps after one warm step
            s1 = new_sched(5)
            st1 = make_stepper(wl, s1)
            st1(0)
            torch.cuda.synchronize()
            ts1 = []
            for i in range(3):
                t0 = time.perf_counter()
                st1(1 + i)
                torch.cuda.synchronize()
                ts1.append(time.perf_counter() - t0)
            single_same = round(1.0 / statistics.median(ts1), 4)
            if verify is not None and lat0_verify is not None:
                # the SAME step (same seeded inputs, fresh scheduler) unsharded on this GPU: what the sharded answer is checked against.
                # rel-L2 of the latents the step leaves, and of the update alone (x_new - x_old: the part the forward decides)
                wv, l0, lat_sh = lat0_verify
                wv.latents.copy_(l0)
                make_stepper(wv, new_sched(3), sequential=a.sequential_cfg)(0)
                torch.cuda.synchronize()
                ref, got, base = wv.latents.double(), lat_sh.double(), l0.double()
                verify["sharded_vs_single_rel_l2"] = float((got - ref).norm() / ref.norm())
                verify["sharded_vs_single_rel_l2_of_update"] = float(((got - base) - (ref - base)).norm() / (ref - base).norm())
                verify["what"] = ("one guidance step from the same seeded latents: sharded over the ranks vs unsharded on rank 0's GPU; bf16 kernels "
                                  "on both sides (different summation orders: GEMM M split, attention key-tile order): expect <= 1e-2 on the update")
        wl2 = Workload(dev, 2, h, w, 42 + rank)
        s2 = new_sched(4)
        dt2 = timed(make_stepper(wl2, s2), 1, 2)
        replica = {"value": round(2 / dt2 * world, 4), "unit": "denoising-steps/sec, aggregate of independent configs[1] edits (one per GPU)",
                   "scaling": "weak", "ms_per_step": round(dt2 / 2 * 1e3, 2)}
        model._sp, model._cfgp = sp, cfgp
        model.engine()._ws = {}

    if rank == 0:
        vae_s, enc_s = single["vae_s"], single["enc_s"]
        steps_per_s = a.steps / dt * (1 if ulysses else world)
        fl = dit_flops_per_forward(N, num_layers=a.layers) * fwd_per_step
        per_gpu = fl * a.steps / dt / 1e12 / (world if ulysses else 1)
        out = {
            "metric": "denoising-steps/sec", "value": round(steps_per_s, 4),
            "unit": f"denoising-steps/sec (ChronoEdit-14B, {a.width}x{a.height}; steps {a.warmup}..{a.warmup + a.steps - 1} of a {max(50, total)}-step flow-UniPC schedule timed)" + (
                f"; ONE configs[3] edit (N = {N} tokens) sharded over {world} GPUs - a different workload from the N = 1 line (configs[1], "
                "N = 7200): read the curve through strong_scaling_speedup_vs_one_gpu" if ulysses else ""),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong" if ulysses else "weak", "vs_baseline": None,
            # from the path actually taken (transformer.attention_path()): under sequence parallelism the self-attention is the bf16 kernel
            "dtype": (("fp8 e4m3 GEMMs (" + ("per-row scales" if a.fp8_row_scales else "OCP-MX block scales, applied in the matrix pipe") + "; fp32 accumulate), bf16 attention / norms / residual"
                       if model.attention_path() == "bf16" else
                       f"fp8 (policy {a.fp8_policy}): e4m3 GEMMs (" + ("per-row scales" if a.fp8_row_scales else "OCP-MX block scales") + ") + MXFP8 self-attention on the MX matrix instruction "
                       "(fp32 accumulate); bf16 cross-attention / norms / residual"))
                     if a.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"ChronoEdit-14B DiT ({a.layers} blocks), {a.width}x{a.height}, {T} latent frames (N={N} tokens), "
                                   f"guidance {a.guidance} ({fwd_per_step} forwards/step) + CFG + flow-UniPC update; "
                                   + _baseline_config_name(a, T),
                       "tokens": N, "forwards_per_step": fwd_per_step,
                       "parallelism": (f"ulysses sp{model._sp.world}" + (" x cfg2" if a.cfg_parallel else "")) if ulysses else f"replica x{world}",
                       "cfg": ("parallel (two Ulysses groups)" if a.cfg_parallel else
                               "sequential (2 x B=1)" if a.sequential_cfg else "batched (1 x B=2) inside the Ulysses group: blocked receive layout") if ulysses
                              else ("sequential (2 x B=1)" if a.sequential_cfg else "batched (1 x B=2)"),
                       "context_cache": bool(a.cache_context)},
            "model_tflops_per_step": round(fl / 1e12, 2),
            "achieved_tflops_per_gpu": round(per_gpu, 1),
            # against the dense peak of the arithmetic this run computes its GEMMs in (fp8 runs: the 5 PFLOP/s fp8 peak, not the bf16 one)
            "mfma_roofline_frac_whole_step": round(per_gpu / (PEAK_FP8_TFLOPS if a.fp8 else PEAK_BF16_TFLOPS), 4),
            "mfma_roofline_peak_tflops": PEAK_FP8_TFLOPS if a.fp8 else PEAK_BF16_TFLOPS,
            "finite": finite,
            **({"TEST_ONLY": f"ranks share one GPU, collectives host-staged over {test_backend}: exercises the code path, measures nothing"} if test_backend else {}),
            **({"TEST_ONLY": "CE_BENCH_ONE_RANK_SP: the sharded code path on a Ulysses group of ONE rank over real RCCL - exercises the path, measures nothing"} if one_rank_sp else {}),
            "launch": "hipGraph replay" if a.graph else "eager",
            "host_enqueue_ms_per_step": host_enqueue_ms,
            "rccl": rccl,
            "sharded_vs_single_rel_l2": None if not verify else verify.get("sharded_vs_single_rel_l2"),
            "sharded_verification": verify,
            **({"sharded_error": sharded_error, "fallback": "the sharded leg failed; this line is the --parallel replica measurement"} if sharded_error else {}),
            "single_gpu_same_workload_steps_per_sec": single_same,
            "strong_scaling_speedup_vs_one_gpu": None if not single_same else round(steps_per_s / single_same, 3),
            "replica_mode": replica,
            "steps_per_sec_with_context_kv_cache": single["cached_rate"],
            "steps_per_sec_fp8_mode": single["fp8_rate"],
            "fp8_mode_frac_of_fp8_peak": None if not single["fp8_rate"] else round(fl * single["fp8_rate"] / 1e12 / PEAK_FP8_TFLOPS, 4),
            "vae": vae_s,
            "encoders": enc_s,
            "sec_per_edit": {"configs[2] 8-step distilled schedule, guidance 1 (measured end to end)": single["edit8"],
                             "configs[1] 50 steps, guidance 5 (measured end to end)": single["edit50"],
                             "configs[1] 50 steps, composed = 50 x ms_per_step + VAE + encoders": None if vae_s is None else round(
                                 50 * dt / a.steps + vae_s["encode_s"] + vae_s["decode_s"] + (0.0 if enc_s is None else enc_s["text_s"] + enc_s["image_s"]), 2)},
            "power": power,
            "roofline": roofline,
            "roofline_family": roofline_family,
            "kernel_breakdown": breakdown,
            "sec_per_edit_temporal_reasoning": single.get("edit_reasoning") if world == 1 else sharded_edit,
            "steps_per_sec_fp8_config4": single.get("fp8_config4"),
            "fp8_policies": single.get("fp8_policies"),
        }
        if a.layers != 40:
            out["invalid"] = "reduced depth (debug run)"
        if not a.no_cpu_baseline:
            try:  # (rank 0 at every N, after the timed region; at N = 28 800 one call of the block is ~1 min of host time: a bounded single-run sample)
                out["cpu_baseline"] = cpu_baseline(N, fwd_per_step, grid=(T, h // 2, w // 2))
            except Exception as e:  # the baseline must never take the bench line down
                out["cpu_baseline"] = {"error": repr(e)}
            try:  # BASELINE.json configs[0] (N = 512, fp32 CPU plumbing): the reference's own ChronoEditTransformer3DModel and the port, depths 1 and 2
                if world == 1:
                    out["cpu_config0"] = cpu_config0(with_reference=True, depths=(1, 2))  # (1, 2, 4): bench.py --cpu-only
            except Exception as e:
                out["cpu_config0"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or one_rank_sp:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 - a data group that failed may not shut down cleanly; the lines are already printed
            pass


def _edit_runners(a, model, vae, te_model, ie_model, dev, after=None):
    """Closures that run ONE whole edit through ChronoEditPipeline (token ids + pixel values + image in, video out) and return its wall time;
    `after(seconds) -> seconds` lets the N > 1 caller take the MAX over ranks.  ONE pipeline object for all edits, as a serving process holds
    it: hipGraph replay (where the step is capturable) and the per-edit context cache are its defaults, and a latent shape seen before is
    captured without another warm-up step.  te_model / ie_model None: seeded random conditioning stands in for the encoders (said in the line)."""
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    g = torch.Generator(device=dev).manual_seed(43)
    image = torch.rand((1, 3, a.height, a.width), generator=g, device=dev) * 2 - 1
    ids = torch.randint(2, 256384, (1, 512), generator=g, device=dev)
    am = torch.zeros((1, 512), dtype=torch.long, device=dev)
    am[0, :64] = 1
    nids = torch.randint(2, 256384, (1, 512), generator=g, device=dev)
    nam = torch.zeros((1, 512), dtype=torch.long, device=dev)
    nam[0, :20] = 1
    px = torch.randn((1, 3, 224, 224), generator=g, device=dev)
    fake = None
    if te_model is None or ie_model is None:
        wl_ = Workload(dev, 2, 16, 16, 4343)
        fake = (wl_.prompt, wl_.negative, wl_.image)
    after = after or (lambda x: x)

    pipe = ChronoEditPipeline(text_encoder=te_model, image_encoder=ie_model, transformer=model, vae=vae,
                              scheduler=FlowUniPCMultistepScheduler(flow_shift=5.0))

    def conditioning(guidance):
        if fake is not None:
            return fake[0], (fake[1] if guidance > 1 else None), fake[2]
        pos, neg = pipe.encode_prompt(input_ids=ids, attention_mask=am, negative_input_ids=nids if guidance > 1 else None,
                                      negative_attention_mask=nam if guidance > 1 else None)
        return pos, neg, pipe.encode_image(px)

    def edit(steps, guidance, shift):
        pipe.scheduler = FlowUniPCMultistepScheduler(flow_shift=shift)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pos, neg, img = conditioning(guidance)
        video = pipe.edit_tensors(image, pos, neg, img, num_frames=5, num_inference_steps=steps, guidance_scale=guidance)
        torch.cuda.synchronize()
        return round(after(time.perf_counter() - t0), 3), bool(torch.isfinite(video.float()).all().item())

    def reasoning_edit(steps, rsteps):
        """BASELINE configs[3], end to end: 29 pixel frames -> 8 latent frames for the first `rsteps` steps, truncated to 2 latent
        frames for the rest (pipeline_chronoedit.py:700-709), both decodes (:776-779)."""
        pipe.scheduler = FlowUniPCMultistepScheduler(flow_shift=5.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pos, neg, img = conditioning(5.0)
        video = pipe.edit_tensors(image, pos, neg, img, num_frames=29, num_inference_steps=steps, guidance_scale=5.0,
                                  enable_temporal_reasoning=True, num_temporal_reasoning_steps=rsteps)
        torch.cuda.synchronize()
        return round(after(time.perf_counter() - t0), 2), tuple(video.shape), bool(torch.isfinite(video.float()).all().item())

    return edit, reasoning_edit


def _sharded_edit_leg(a, model, dev, world, after, test_backend):
    """N > 1: WHOLE temporal-reasoning edits (BASELINE configs[3]) with the DiT sharded over the ranks - run by EVERY rank, outside the timed
    region, MAX over ranks of the wall time.  What is and is not sharded: the DiT forwards are (token axis over the Ulysses group, or the
    guidance pair over two groups); the VAE encode of the 29 frames, the two VAE decodes and the UMT5 / CLIP encoders run REPLICATED on every
    rank (each GPU computes the whole of them - the causal temporal cache serialises the VAE, SURVEY section 8e - so they are the serial
    fraction of the edit), and every rank ends with the full video.  Under CE_BENCH_TEST_BACKEND the schedule is 3 steps (the code path, not a
    measurement)."""
    from chronoedit_amd.vae import AutoencoderKLWan
    vae = AutoencoderKLWan.random_init(dev, seed=4321)
    te_model = ie_model = None
    if not a.no_encoders:
        from chronoedit_amd.clip_vision import CLIPVisionModel
        from chronoedit_amd.umt5 import UMT5EncoderModel
        torch.manual_seed(0)
        te_model, ie_model = UMT5EncoderModel(device=dev), CLIPVisionModel(device=dev)
    _, reasoning_edit = _edit_runners(a, model, vae, te_model, ie_model, dev, after=after)
    steps = 3 if test_backend else 50
    out = {}
    for rs in [int(x) for x in str(a.reasoning_steps).split(",") if x.strip()]:
        rs = max(0, min(steps, rs if not test_backend else (1 if rs < 50 else steps)))
        sr, shape_r, okr = reasoning_edit(steps, rs)
        out[f"num_temporal_reasoning_steps={rs}" + (" (the reference's default: never truncates)" if rs >= steps else "")] = {
            "seconds": sr, "finite": okr, "frames": shape_r[2], "num_temporal_reasoning_steps": rs, "num_inference_steps": steps, "n_gpus": world,
            "includes": ("UMT5 (2 prompts) + CLIP" if te_model is not None else "(encoders skipped: seeded random conditioning)") +
                        f" + VAE encode of 29 frames + {rs} steps x 2 forwards with 8 latent frames + {steps - rs} steps x 2 forwards with 2 latent frames + two VAE "
                        "decodes; MAX over ranks of the wall time",
            "sharding": "DiT forwards sharded over the ranks (as in `config.parallelism`); VAE encode / decodes and the UMT5 / CLIP encoders replicated on every rank "
                        "(the serial fraction); launch " + ("eager (the exchanges are torch.distributed collectives)" if not getattr(getattr(model, "_sp", None), "capturable", False)
                                                            else "hipGraph replay on the library-owned communicator")}
    del vae, te_model, ie_model
    return out


def _single_gpu_secondaries(a, model, wl, new_sched, make_stepper, dev, T, h, w, out):
    """Secondary figures of the one-GPU run, all outside the timed region."""
    base_sched = new_sched(12)
    step = make_stepper(wl, base_sched, sequential=a.sequential_cfg)
    # the same step with the step-invariant text / image K/V projections computed once and reused (identical
    # results, 1.4 % fewer flops) - reported beside `value`, never as it
    if not a.cache_context and not a.graph:
        model.cache_context = True
        step(0)  # fills the cache
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(2):
            step(1 + i)
        torch.cuda.synchronize()
        out["cached_rate"] = round(2 / (time.perf_counter() - tc), 4)
        model.cache_context = False
        model.clear_context_cache()
    # the same step in the fp8 GEMM mode (BASELINE.json configs[4] arithmetic) - a different precision,
    # reported beside the bf16 `value`, never as it
    if not a.fp8 and not a.graph and not a.no_fp8_leg:
        model.enable_fp8_gemms()
        model.enable_fp8_attention()
        step(4)  # packs the e4m3 weights
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(10):  # (ten timed steps after the warm one: VERDICT r5 - two were too few to quote)
            step(5 + i)
        torch.cuda.synchronize()
        out["fp8_rate"] = round(10 / (time.perf_counter() - tc), 4)
        # BASELINE.json configs[4] itself: the same fp8 arithmetic at the upscaler shape, 1584x1056 px -> latents [1,16,2,132,198] -> N = 13 068 tokens
        # (README.md:149-158), guidance 5: one warm step (new workspaces), then ten timed; fraction against the FP8 peak
        if (a.width, a.height, T) == (1280, 720, 2) and not a.no_fp8_config4:
            from chronoedit_amd.flops import dit_flops_per_forward
            wl4 = Workload(dev, 2, 1056 // 8, 1584 // 8, 44)
            st4 = make_stepper(wl4, new_sched(12), sequential=a.sequential_cfg)
            st4(0)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for i in range(10):
                st4(1 + i)
            torch.cuda.synchronize()
            d4 = (time.perf_counter() - tc) / 10
            fl4 = dit_flops_per_forward(wl4.N, num_layers=a.layers) * 2
            out["fp8_config4"] = {"value": round(1.0 / d4, 4), "unit": "denoising-steps/sec", "ms_per_step": round(d4 * 1e3, 2), "steps": 10, "warmup": 1,
                                  "workload": f"BASELINE.json configs[4]: 1584x1056, 2 latent frames (N = {wl4.N} tokens), guidance 5 (2 forwards/step batched), fp8 e4m3 GEMMs "
                                              "(OCP-MX block scales) + MXFP8 self-attention, eager, nothing cached",
                                  "model_tflops_per_step": round(fl4 / 1e12, 2), "achieved_tflops": round(fl4 / d4 / 1e12, 1),
                                  "frac_of_fp8_peak": round(fl4 / d4 / 1e12 / PEAK_FP8_TFLOPS, 4), "peak": PEAK_FP8_TFLOPS,
                                  "finite": bool(torch.isfinite(wl4.latents).all().item())}
            del wl4, st4
        # the mixed-precision policy beside it (round 6): "fp8-accurate" = the ungated cross-attention out-projection back on the bf16 GEMM.  The
        # error ratios are the committed full-width measurement (one block at N = 7 200 against the fp32 oracle: profiles/r06_fp8_sensitivity.txt,
        # tools/fp8_sensitivity.py; pinned by tests/test_bench_shapes_gpu.py), the rates are measured here
        model.enable_fp8_gemms(policy="accurate")
        step(15)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for i in range(10):
            step(16 + i)
        torch.cuda.synchronize()
        acc_rate = round(10 / (time.perf_counter() - tc), 4)
        out["fp8_policies"] = {
            "fp8-fast": {"steps_per_sec": out["fp8_rate"], "linears_in_fp8": list(model.FP8_POLICIES["fast"]), "self_attention": "MXFP8",
                         "block_error_vs_fp32_over_bf16_path": 7.85},
            "fp8-accurate": {"steps_per_sec": acc_rate, "linears_in_fp8": list(model.FP8_POLICIES["accurate"]), "self_attention": "MXFP8",
                             "block_error_vs_fp32_over_bf16_path": 4.0},
            "bf16": {"block_error_vs_fp32": 4.76e-3},
            "error_source": "profiles/r06_fp8_sensitivity.txt (one full-width block, N = 7200, synthetic weights: the AdaLN gates are small, which favours the gated Linears)"}
        model.enable_fp8_gemms(False)
        model.enable_fp8_attention(False)
    # VAE encode + decode at the same resolution (once per edit)
    vae = None
    if not a.no_vae:
        from chronoedit_amd.vae import AutoencoderKLWan
        vae = AutoencoderKLWan.random_init(dev, seed=4321)
        nf = 4 * (T - 1) + 1
        vid = (torch.rand(1, 3, nf, a.height, a.width, device=dev) * 2 - 1).to(torch.bfloat16)
        zl = torch.randn(1, 16, T, h, w, device=dev).to(torch.bfloat16)
        def vae_pair():
            torch.cuda.synchronize()
            tv = time.perf_counter()
            vae.encode(vid).latent_dist.mode()
            torch.cuda.synchronize()
            te = time.perf_counter() - tv
            tv = time.perf_counter()
            vae.decode(zl, return_dict=False)
            torch.cuda.synchronize()
            return round(te, 4), round(time.perf_counter() - tv, 4)

        vae.encode(vid), vae.decode(zl)  # warm-up
        eager = vae_pair()
        # the product pipeline's default (ChronoEditPipeline.use_graph): from the second edit of a shape on, encode and decode are one
        # hipGraph replay each - same kernels, no allocator traffic between them
        vae.use_graph = True
        vae.encode(vid), vae.decode(zl)  # (first call of a shape under use_graph: eager)
        vae.encode(vid), vae.decode(zl)  # capture + replay
        replay = vae_pair()
        out["vae_s"] = {"encode_s": replay[0], "decode_s": replay[1], "launch": "one hipGraph replay each (pipeline default)",
                        "eager_encode_s": eager[0], "eager_decode_s": eager[1]}
    # conditioning encoders (once per edit): UMT5-XXL on the positive + negative prompt padded to 512 tokens, CLIP ViT-H/14
    te_model = ie_model = None
    if not a.no_encoders:
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
        out["enc_s"] = {"text_s": round(tt_, 4), "image_s": round(time.perf_counter() - tv, 4),
                        "finite": bool(torch.isfinite(pe.float()).all().item() and torch.isfinite(ie.float()).all().item())}
    # MEASURED sec/edit through ChronoEditPipeline: token ids + pixel values + image in, video out
    if vae is not None and te_model is not None and not a.no_edit and (a.width, a.height, T) == (1280, 720, 2):
        edit, reasoning_edit = _edit_runners(a, model, vae, te_model, ie_model, dev)

        edit(2, 1.0, 2.0)  # warm-up of the B = 1 shapes
        s8, ok8 = edit(8, 1.0, 2.0)
        out["edit8"] = {"seconds": s8, "finite": ok8, "includes": "UMT5 (1 prompt) + CLIP + VAE encode + 8 steps x 1 forward + VAE decode; pipeline defaults: "
                                                                  "one hipGraph replay per step (shape seen before: no warm-up step), context projections once per edit"}
        if a.full_edit:
            s50, ok50 = edit(50, 5.0, 5.0)
            out["edit50"] = {"seconds": s50, "finite": ok50, "includes": "UMT5 (2 prompts) + CLIP + VAE encode + 50 steps x 2 forwards + VAE decode; "
                                                                       "one hipGraph replay per step, context projections once per edit"}
        if a.reasoning_edit:
            out["edit_reasoning"] = {}
            for rs in [int(x) for x in str(a.reasoning_steps).split(",") if x.strip()]:
                rs = max(0, min(50, rs))
                sr, shape_r, okr = reasoning_edit(50, rs)
                out["edit_reasoning"][f"num_temporal_reasoning_steps={rs}" + (" (the reference's default: never truncates)" if rs >= 50 else "")] = {
                    "seconds": sr, "finite": okr, "frames": shape_r[2], "num_temporal_reasoning_steps": rs,
                    "includes": f"UMT5 (2 prompts) + CLIP + VAE encode of 29 frames + {rs} steps x 2 forwards at N = 28800 + "
                                f"{50 - rs} steps x 2 forwards at N = 7200 + two VAE decodes; hipGraph replay, context projections once per edit"}


if __name__ == "__main__":
    main()
