"""Denoising loop of ChronoEditPipeline.__call__ on the HIP engine
(/root/reference/chronoedit_diffusers/pipeline_chronoedit.py:694-756).

`denoise_step` is one loop iteration — the unit BASELINE.json's "denoising-steps/sec" counts:
cat(latents, condition) -> 1 or 2 DiT forwards -> CFG -> scheduler update.  The latents stay fp32
on the device for the whole trajectory; the CFG combine and the UniPC update are one fused launch.
"""
from __future__ import annotations

import html
import re
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch

from .scheduler import FlowUniPCMultistepScheduler
from .transformer import ChronoEditTransformer3DModel


def _is_pil(x) -> bool:
    try:
        from PIL import Image
    except ImportError:  # pragma: no cover
        return False
    return isinstance(x, Image.Image)


def prompt_clean(text: str) -> str:
    """pipeline_chronoedit.py:98-112: ftfy.fix_text (when ftfy is installed, as in the reference), double html.unescape,
    whitespace collapse."""
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


def make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds):
    """[cond | uncond] conditioning stacked along the batch axis; build it ONCE per edit so that the transformer's
    context cache (step-invariant text/image K/V) can recognise the tensors across steps."""
    text2 = torch.cat([prompt_embeds, negative_prompt_embeds], 0)
    image2 = None if image_embeds is None else torch.cat([image_embeds, image_embeds], 0)
    return text2, image2


def _token_sharded(transformer) -> bool:
    sp = getattr(transformer, "_sp", None)
    return sp is not None and sp.sharded


def _capturable(transformer) -> bool:
    """Can a step of this transformer be captured into a hipGraph?  Unsharded: yes.  Token-sharded: only when the exchanges run on the
    library-owned RCCL communicator (`enable_sequence_parallel(owned_comm=True)`, parallel.OwnedComm); CFG parallelism (its pairwise
    exchange is a torch.distributed collective) runs eagerly."""
    if getattr(transformer, "_cfgp", None) is not None:
        return False
    if not _token_sharded(transformer):
        return True
    return bool(getattr(transformer._sp, "capturable", False))


def _sharded_batchable(transformer) -> bool:
    """Can a token-sharded (Ulysses) forward take the guidance pair as ONE batch of two?  Yes on the V^T attention path (the
    blocked-layout kernels, chronoedit_amd/parallel.py) - with bf16 or fp8 GEMMs; the sharded self-attention itself is always the
    bf16 kernel (transformer.attention_path()).  The register-staged attention kernel runs the passes in sequence."""
    return bool(getattr(transformer, "v_transposed", False) and getattr(transformer, "sp_batch_cfg", True))


@torch.no_grad()
def denoise_step(transformer: ChronoEditTransformer3DModel, scheduler: FlowUniPCMultistepScheduler, latents: torch.Tensor,
                 condition: torch.Tensor, t: torch.Tensor, prompt_embeds: torch.Tensor,
                 negative_prompt_embeds: Optional[torch.Tensor], image_embeds: Optional[torch.Tensor],
                 guidance_scale: float, batch_cfg: bool = True, cfg_inputs=None) -> torch.Tensor:
    """latents fp32 [B,16,T,h,w] (updated in place), condition bf16 [B,20,T,h,w]  (pipeline_chronoedit.py:711-739)."""
    latent_model_input = torch.cat([latents.to(torch.bfloat16), condition], dim=1)
    B = latents.shape[0]
    timestep = t.expand(B)
    batch_cfg = batch_cfg and (not _token_sharded(transformer) or _sharded_batchable(transformer))
    cfgp = getattr(transformer, "_cfgp", None)
    if guidance_scale > 1.0 and negative_prompt_embeds is not None:  # do_classifier_free_guidance
        if cfgp is not None:
            # CFG parallelism: this rank's Ulysses group runs ONE of the two passes, the predictions are exchanged pairwise
            text = prompt_embeds if cfgp.branch == 0 else negative_prompt_embeds
            mine = transformer(latent_model_input, timestep, text, image_embeds, return_dict=False)[0]
            noise_pred, noise_uncond = cfgp.exchange(mine)
        elif batch_cfg:
            # the conditional and unconditional passes (pipeline_chronoedit.py:715-735) as ONE forward over 2B samples:
            # identical per-sample arithmetic, but every weight streams from HBM once and the GEMM grids fill the chip
            text2, image2 = cfg_inputs if cfg_inputs is not None else make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
            out = transformer(torch.cat([latent_model_input, latent_model_input], 0), torch.cat([timestep, timestep], 0),
                              text2, image2, return_dict=False)[0]
            noise_pred, noise_uncond = out[:B].contiguous(), out[B:].contiguous()
        else:
            noise_pred = transformer(latent_model_input, timestep, prompt_embeds, image_embeds, return_dict=False)[0]
            noise_uncond = transformer(latent_model_input, timestep, negative_prompt_embeds, image_embeds, return_dict=False)[0]
    else:
        noise_pred = transformer(latent_model_input, timestep, prompt_embeds, image_embeds, return_dict=False)[0]
        noise_uncond = None
    return scheduler.step_cfg(noise_pred, noise_uncond, guidance_scale, latents)


_WARMUP_STREAMS = {}


def _warmup_stream(dev: torch.device) -> "torch.cuda.Stream":
    """The side stream the un-captured step in front of a capture runs on - one per device for the life of the process (a fresh stream per
    GraphedDenoiser left a 64 MiB split-K scratch registered for each of them: ADVICE r4)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _WARMUP_STREAMS.get(key)
    if st is None:
        st = _WARMUP_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class GraphedDenoiser:
    """One denoising step captured as a hipGraph and replayed per step (north star: "the 50-step / 8-step sampling loops
    are hipGraph-captured").  Everything a step reads lives at fixed device addresses: the fp32 latents (updated in place),
    the condition, the stacked CFG conditioning, the scheduler history, and two small staging buffers that receive the
    step's timestep and UniPC coefficient row (device-to-device copies enqueued in front of the replay - no host sync).
    A new graph is needed when the latent shape changes (temporal-reasoning truncation 8 -> 2 frames).
    `step(i)` raises RuntimeError when the object was built at one step index with keep_warmup_step=True (that step ran eagerly in the
    constructor) and the first `step()` asks for another index: build it at the index it starts from.  Between construction and the last
    `step()` the transformer's modes (fp8, sequence parallelism, cache_context) must not change; other forwards of the same transformer
    are allowed (the lazy capture re-primes its context entry)."""

    def __init__(self, transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
                 guidance_scale: float, batch_cfg: bool = True, warm: bool = False, keep_warmup_step: bool = True):
        """warm: this process has already run a step of exactly this shape / guidance form through `transformer` (packed weights,
        workspaces, kernel attributes exist), so no un-captured step is needed in front of the capture; only the step-invariant context
        projections are computed eagerly so that the graph holds the cache HIT (`cache_context`), not the projections.
        Not warm: the un-captured step that triggers the lazy initialisations IS the trajectory's current step (it runs eagerly on the
        live state, `step()` then skips the replay for that index) - a discarded warm-up cost a whole step per new shape: 2 s of a
        temporal-reasoning edit at 28 800 tokens, 12 % of an 8-step edit.  keep_warmup_step=False: run it on saved state and put the
        state back (a caller that wants every `step()` to be a replay: bench.py)."""
        assert latents.dtype == torch.float32 and latents.is_contiguous()
        if not _capturable(transformer):
            # Measured on this stack (ROCm 7.0 / RCCL 2.26 / torch 2.10: tools/rccl_graph_probe.py, profiles/r03_rccl_graph_probe.txt): ONE
            # collective of torch.distributed's "nccl" backend can be captured and replayed, but the process-group watchdog thread keeps
            # polling the event of every Work created under capture - the next capture (or any later poll) kills the process
            # (hipErrorStreamCaptureUnsupported in "global" capture mode, a segmentation fault in "thread_local" / "relaxed"), with
            # synchronous collectives, async_op=True and side-stream fork / join alike.  A sharded loop needs two graphs (8 -> 2 frames), so it
            # runs eagerly on torch's communicator; DESIGN.md section 6 has the measurement of why that costs nothing at the per-rank kernel
            # times of 4 / 8 GPUs.  The way to capture it: the library-owned communicator (`enable_sequence_parallel(owned_comm=True)`).
            raise NotImplementedError("hipGraph capture of a step with torch.distributed exchanges is not usable on this torch / RCCL build: "
                                      "enable_sequence_parallel(owned_comm=True) or run the sharded loop eagerly")
        self.tr, self.sch, self.latents, self.condition = transformer, scheduler, latents, condition
        self.prompt, self.negative, self.image, self.g, self.batch_cfg = prompt_embeds, negative_prompt_embeds, image_embeds, guidance_scale, batch_cfg
        dev = latents.device
        self.cfg_inputs = None
        self.guided = guidance_scale > 1.0 and negative_prompt_embeds is not None
        if self.guided and batch_cfg:
            self.cfg_inputs = make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
        self.t_buf = torch.zeros((), dtype=torch.int64, device=dev)
        self.coef_buf = torch.zeros(10, dtype=torch.float32, device=dev)
        scheduler._ensure_state(latents)
        if scheduler.last_sample is None:
            scheduler.last_sample = torch.zeros_like(latents)
        idx0 = scheduler._step_index
        self._stage(idx0 or 0)
        eng = getattr(transformer, "_engine", None)
        n_samples = latents.shape[0] * (2 if self.cfg_inputs is not None else 1)
        warm = bool(warm and eng is not None and hasattr(eng, "is_warm") and eng.is_warm(n_samples, *latents.shape[2:]))
        self._done_index = None  # the step that already ran eagerly (below); step() does not replay it
        if not warm:
            # lazy initialisations (packed weights, workspaces, function attributes) must not happen under capture: run the CURRENT step
            # eagerly, for real, on a side stream as torch asks of anything that precedes a capture
            saved = None if keep_warmup_step else (latents.clone(), [m.clone() for m in scheduler.model_outputs], scheduler.last_sample.clone())
            side = _warmup_stream(dev)  # ONE long-lived side stream per device: every stream GEMMs run on gets a split-K scratch of its own (ops.ensure_gemm_workspace)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._body()
            torch.cuda.current_stream().wait_stream(side)
            if keep_warmup_step:
                self._done_index = idx0 or 0
            else:
                self.latents.copy_(saved[0])
                for m, sv in zip(scheduler.model_outputs, saved[1]):
                    m.copy_(sv)
                scheduler.last_sample.copy_(saved[2])
        else:
            self._prime_context()
        # The capture: at once when every step() must be a replay (warm shape, or keep_warmup_step=False); LAZILY - at the first step() that needs a
        # replay - when the constructor already ran the trajectory's current step eagerly: if that was the last step of this shape (truncation lands
        # there, or a callback replaces the latents every step) nothing is ever replayed and a whole capture would be thrown away
        self.graph = None
        if self._done_index is None:
            self._capture()
        scheduler._step_index = idx0  # step_cfg counts on the host, also while being captured

    def _prime_context(self):
        """With `cache_context`: the step-invariant context projections of THIS conditioning computed eagerly now, so that the capture
        records the cache hit and not the projections."""
        tr = self.tr
        if getattr(tr, "cache_context", False) and hasattr(tr, "prime_context"):
            if self.cfg_inputs is not None:
                tr.prime_context(self.cfg_inputs[0], self.cfg_inputs[1])
            elif not self.guided:
                tr.prime_context(self.prompt, self.image)

    def _capture(self):
        # The LAZY capture (first replayed step()) runs at an arbitrary later point of the caller's program: another forward of the same
        # transformer in between (another conditioning of the same shape) may have replaced the context-cache entry and the engine-owned
        # context buffers the eager step left, so the entry is re-primed here (a hit costs nothing) - ADVICE r5.  torch.cuda.graph() itself
        # synchronises the device and runs a gc pass: callers that cannot afford that inside their loop build with keep_warmup_step=False
        # (capture in the constructor).
        self._prime_context()
        idx = self.sch._step_index
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):  # (a capture executes nothing: the device state stays what the eager step left)
            self._body()
        self.sch._step_index = idx

    def _stage(self, i):
        self.t_buf.copy_(self.sch.timesteps[i])
        self.coef_buf.copy_(self.sch.coef_row(i, self.g, self.latents.device))

    def _body(self):
        inp = torch.cat([self.latents.to(torch.bfloat16), self.condition], dim=1)
        B = inp.shape[0]
        ts = self.t_buf.expand(B)
        if self.cfg_inputs is not None:
            out = self.tr(torch.cat([inp, inp], 0), torch.cat([ts, ts], 0), self.cfg_inputs[0], self.cfg_inputs[1], return_dict=False)[0]
            c, u = out[:B].contiguous(), out[B:].contiguous()
        elif self.guided:
            c = self.tr(inp, ts, self.prompt, self.image, return_dict=False)[0]
            u = self.tr(inp, ts, self.negative, self.image, return_dict=False)[0]
        else:
            c, u = self.tr(inp, ts, self.prompt, self.image, return_dict=False)[0], None
        self.sch.step_cfg(c, u, self.g, self.latents, coef=self.coef_buf)

    def step(self, i: int) -> torch.Tensor:
        if self._done_index is not None:  # the constructor already ran ONE real step (index _done_index) eagerly on the live state
            if i != self._done_index:
                # a caller that re-seeded the latents or starts elsewhere would silently get a trajectory with one extra step
                raise RuntimeError(f"GraphedDenoiser was built at step {self._done_index} (which it ran eagerly, keep_warmup_step=True) but step({i}) "
                                   "was asked first: build it at the step it starts from, or with keep_warmup_step=False")
            self._done_index = None
            self.sch._step_index = i + 1
            return self.latents
        if self.graph is None:
            self._capture()
        self._stage(i)
        self.graph.replay()
        self.sch._step_index = i + 1
        return self.latents


@torch.no_grad()
def denoise(transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
            num_inference_steps: int, guidance_scale: float = 5.0, enable_temporal_reasoning: bool = False,
            num_temporal_reasoning_steps: int = 0, use_graph: bool = False, on_step_end=None, interrupted=None, graph_warm=None):
    """The whole loop, including the temporal-reasoning truncation 8 -> 2 latent frames (pipeline_chronoedit.py:700-709).
    on_step_end(i, t, latents) -> replacement latents, a dict with any of latents / prompt_embeds / negative_prompt_embeds, or None
    (the reference's callback_on_step_end hook, :741-749); graph_warm: a set the caller keeps across edits - shapes already run once in
    this process skip GraphedDenoiser's warm-up step;
    interrupted() -> True skips the remaining steps (`self.interrupt`, :697-698).  With the tokens sharded over ranks
    (Ulysses) every rank holds the replicated latents and scheduler history, so the truncation is a local slice on every
    rank (the reference all-gathers and re-shards: chronoedit_14b_edit_model.py:168-186) and the next forward simply shards
    the shorter sequence."""
    scheduler.set_timesteps(num_inference_steps, device=latents.device)
    latents = latents.to(torch.float32).contiguous()
    sharded = getattr(transformer, "_cfgp", None) is not None or (_token_sharded(transformer) and not _sharded_batchable(transformer))
    cfg_inputs = None
    if guidance_scale > 1.0 and negative_prompt_embeds is not None and not sharded:
        cfg_inputs = make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
    if hasattr(transformer, "clear_context_cache"):
        transformer.clear_context_cache()  # a new edit: nothing of the previous edit's conditioning may be reused
    graphed = None
    if use_graph and not _capturable(transformer):
        use_graph = False  # a step with torch.distributed exchanges runs eagerly (GraphedDenoiser says why); the pipeline default stays use_graph=True
    for i, t in enumerate(scheduler.timesteps):
        if interrupted is not None and interrupted():
            continue
        if enable_temporal_reasoning and i == num_temporal_reasoning_steps:
            graphed = None  # new latent shape -> new graph
            latents = latents[:, :, [0, -1]].contiguous()
            condition = condition[:, :, [0, -1]].contiguous()
            for j in range(len(scheduler.model_outputs)):
                mo = scheduler.model_outputs[j]
                if mo is not None and mo.shape[-3] != latents.shape[-3]:
                    scheduler.model_outputs[j] = mo[:, :, [0, -1]].contiguous()
            if scheduler.last_sample is not None and scheduler.last_sample.shape[-3] != latents.shape[-3]:
                scheduler.last_sample = scheduler.last_sample[:, :, [0, -1]].contiguous()
        if use_graph:
            if graphed is None:
                if scheduler._step_index is None:
                    scheduler._step_index = i
                # engine_generation(): bumped whenever the transformer's engine, its workspaces or a mode that changes the launch sequence
                # is replaced - a recycled id() or a mode switch that keeps the workspace key must not skip the eager warm-up step
                def warm_key():
                    gen = transformer.engine_generation() if hasattr(transformer, "engine_generation") else id(transformer)
                    # (the scheduler object is not part of the key: everything a step reads from it is created eagerly by GraphedDenoiser
                    # before the capture - _ensure_state, the staged timestep / coefficient row - so a fresh scheduler per edit stays warm)
                    return (tuple(latents.shape), guidance_scale > 1.0 and negative_prompt_embeds is not None, gen)
                graphed = GraphedDenoiser(transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds,
                                          image_embeds, guidance_scale, batch_cfg=not sharded, warm=graph_warm is not None and warm_key() in graph_warm)
                if graph_warm is not None:
                    graph_warm.add(warm_key())  # (taken AFTER the construction: the first step of a process creates the engine)
            latents = graphed.step(i)
        else:
            latents = denoise_step(transformer, scheduler, latents, condition, t, prompt_embeds, negative_prompt_embeds,
                                   image_embeds, guidance_scale, batch_cfg=not sharded, cfg_inputs=cfg_inputs)
        if on_step_end is not None:
            new = on_step_end(i, t, latents)
            if isinstance(new, dict):  # the reference's callback may also replace the conditioning (pipeline_chronoedit.py:747-749)
                if "prompt_embeds" in new or "negative_prompt_embeds" in new:
                    prompt_embeds = new.get("prompt_embeds", prompt_embeds)
                    if negative_prompt_embeds is not None:
                        negative_prompt_embeds = new.get("negative_prompt_embeds", negative_prompt_embeds)
                    if cfg_inputs is not None:
                        cfg_inputs = make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
                    graphed = None  # the graph holds the stacked conditioning it was captured with
                new = new.get("latents")
            if new is not None and new is not latents:
                graphed = None  # the graph is tied to the latents' storage
                latents = new.to(torch.float32).contiguous()
    return latents


# ---------------------------------------------------------------------------------------------------------------
# The edit: prepare_latents -> denoise -> decode  (pipeline_chronoedit.py:392-456, 694-781)
# ---------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def prepare_latents(vae, image: torch.Tensor, num_frames: int, latents: Optional[torch.Tensor] = None, generator=None):
    """image [B,3,H,W] in [-1,1] -> (latents fp32 [B,16,T,h,w], condition bf16 [B,20,T,h,w]).
    pipeline_chronoedit.py:392-456: the condition video is [image, 0, 0, ...]; its VAE posterior mode is normalised with the
    latent mean/std; a 4-channel first-frame mask is stacked in front."""
    B, _, H, W = image.shape
    dev = image.device
    tds = 2 ** sum(vae.temperal_downsample)
    T = (num_frames - 1) // tds + 1
    h, w = H // 8, W // 8
    z = vae.config.z_dim
    if latents is None:
        latents = torch.randn((B, z, T, h, w), generator=generator, device=dev, dtype=torch.float32)
    else:
        latents = latents.to(device=dev, dtype=torch.float32)
    video = torch.cat([image.unsqueeze(2), image.new_zeros(B, 3, num_frames - 1, H, W)], dim=2).to(torch.bfloat16)
    mean = torch.tensor(vae.config.latents_mean, device=dev, dtype=torch.bfloat16).view(1, z, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(vae.config.latents_std)).to(device=dev, dtype=torch.bfloat16).view(1, z, 1, 1, 1)
    cond = vae.encode(video).latent_dist.mode()
    cond = (cond - mean) * inv_std
    mask = torch.ones(B, 1, num_frames, h, w, device=dev)
    mask[:, :, 1:] = 0
    first = torch.repeat_interleave(mask[:, :, 0:1], dim=2, repeats=tds)
    mask = torch.cat([first, mask[:, :, 1:]], dim=2).view(B, -1, tds, h, w).transpose(1, 2)
    return latents.contiguous(), torch.cat([mask.to(cond.dtype), cond], dim=1).contiguous()


@torch.no_grad()
def decode_latents(vae, latents: torch.Tensor, enable_temporal_reasoning: bool = False, num_temporal_reasoning_steps: int = 0):
    """pipeline_chronoedit.py:765-781: de-normalise, decode; in reasoning mode the edit frames and the reasoning frames are
    decoded separately and concatenated."""
    z = vae.config.z_dim
    latents = latents.to(torch.bfloat16)
    mean = torch.tensor(vae.config.latents_mean, device=latents.device, dtype=latents.dtype).view(1, z, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(vae.config.latents_std)).to(device=latents.device, dtype=latents.dtype).view(1, z, 1, 1, 1)
    latents = latents / inv_std + mean
    if enable_temporal_reasoning and num_temporal_reasoning_steps > 0 and latents.shape[2] > 2:
        edit = vae.decode(latents[:, :, [0, -1]], return_dict=False)[0]
        reason = vae.decode(latents[:, :, :-1], return_dict=False)[0]
        return torch.cat([reason, edit[:, :, 1:]], dim=2)
    return vae.decode(latents, return_dict=False)[0]


@dataclass
class WanPipelineOutput:
    """diffusers.pipelines.wan.pipeline_output.WanPipelineOutput: `.frames` is what callers index
    (run_inference_diffusers.py:441 `.frames[0]`)."""
    frames: Any


def _randn_tensor(shape, generator, device, dtype):
    """diffusers.utils.torch_utils.randn_tensor semantics (pipeline_chronoedit.py:418): the noise is drawn on the GENERATOR's
    device (a CPU generator gives CPU-reproducible noise that is then moved), one generator per sample when a list is given."""
    device = torch.device(device)
    gens = generator if isinstance(generator, (list, tuple)) else [generator] * 1
    if isinstance(generator, (list, tuple)):
        shape1 = (1,) + tuple(shape[1:])
        return torch.cat([_randn_tensor(shape1, g, device, dtype) for g in generator], dim=0)
    g = gens[0]
    gdev = device if g is None else torch.device(g.device)
    if gdev.type != device.type:
        if gdev.type == "cpu":
            return torch.randn(shape, generator=g, device="cpu", dtype=dtype).to(device)
        raise ValueError(f"Cannot generate a {device.type} tensor from a generator of type {gdev.type}.")
    return torch.randn(shape, generator=g, device=device, dtype=dtype)


class ChronoEditPipeline:
    """MI355X drop-in for the reference pipeline of the same name (chronoedit_diffusers/pipeline_chronoedit.py:124-812): the same
    constructor components, `from_pretrained`, `encode_prompt` / `encode_image`, `check_inputs`, `prepare_latents`, LoRA entry
    points and `__call__(image=, prompt=, negative_prompt=, height=, width=, num_frames=, num_inference_steps=, guidance_scale=,
    ..., offload_model=) -> WanPipelineOutput(frames=...)`, so `scripts/run_inference_diffusers.py:428-441` drives it unchanged.
    Every tensor op behind it runs on the HIP kernels (DiT, UniPC+CFG, VAE, UMT5, CLIP drop-ins); the tokenizer and the CLIP
    image processor are the reference's host-side transformers objects and are injected (`tokenizer=`, `image_processor=`).
    Guardrails are out of scope: `text_guardrail_runner` / `video_guardrail_runner` stay None unless the
    caller installs callables, which are then honoured with the reference's error behaviour (:621-629, :784-799)."""

    model_cpu_offload_seq = "text_encoder->image_encoder->transformer->vae"
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, tokenizer=None, text_encoder=None, image_encoder=None, image_processor=None,
                 transformer: Optional[ChronoEditTransformer3DModel] = None, vae=None,
                 scheduler: Optional[FlowUniPCMultistepScheduler] = None, disable_guardrails: bool = True):
        self.tokenizer, self.text_encoder, self.image_encoder, self.image_processor = tokenizer, text_encoder, image_encoder, image_processor
        self.vae, self.transformer, self.scheduler = vae, transformer, scheduler  # chronoedit_amd drop-ins
        tds = getattr(vae, "temperal_downsample", None)
        self.vae_scale_factor_temporal = 2 ** sum(tds) if tds is not None else 4      # :185
        self.vae_scale_factor_spatial = 2 ** len(tds) if tds is not None else 8       # :186
        self.guardrail_enabled = not disable_guardrails
        self.text_guardrail_runner = None
        self.video_guardrail_runner = None
        # Defaults of the product pipeline (bit-identical to the eager / uncached loop: tests/test_pipeline_gpu.py): every step of the
        # loop is ONE hipGraph replay (GraphedDenoiser; two graphs when temporal reasoning truncates 8 -> 2 frames), and the
        # step-invariant text / image context projections are computed once per edit (`denoise` clears them at
        # the start of every edit).  `pipe.use_graph = False` / `pipe.transformer.cache_context = False` switch either off.  The VAE
        # follows `use_graph`: from the second edit of a shape on its encode / decode are one graph replay each.
        self.use_graph = True
        self._graph_warm = set()  # (latent shape, guided) pairs whose lazy initialisations have already happened in this process
        if transformer is not None and hasattr(transformer, "cache_context"):
            transformer.cache_context = True
        self._guidance_scale, self._attention_kwargs, self._current_timestep, self._interrupt, self._num_timesteps = 1.0, None, None, False, 0

    # -- properties of the reference pipeline (:458-478) ---------------------------------------------------------------
    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def current_timestep(self):
        return self._current_timestep

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def attention_kwargs(self):
        return self._attention_kwargs

    @property
    def _execution_device(self) -> torch.device:
        for m in (self.transformer, self.vae, self.text_encoder, self.image_encoder):
            d = getattr(m, "device", None)
            if d is not None:
                return torch.device(d)
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def to(self, device=None, dtype=None):
        """`pipe.to(device)` (run_inference_diffusers.py:386): moves every nn.Module component.  The engine computes in bf16 (the
        reference runner hard-codes it, :344-362): any other `dtype` is refused rather than ignored."""
        if dtype is not None and dtype != torch.bfloat16:
            raise ValueError(f"chronoedit_amd components compute in torch.bfloat16; pipe.to(dtype={dtype}) is not supported")
        for m in (self.text_encoder, self.image_encoder, self.transformer, self.vae):
            if m is not None and hasattr(m, "to") and device is not None:
                m.to(device)
        return self

    def maybe_free_model_hooks(self):
        return None

    @classmethod
    def from_pretrained(cls, path: str, transformer=None, vae=None, text_encoder=None, image_encoder=None, scheduler=None,
                        tokenizer=None, image_processor=None, torch_dtype: torch.dtype = torch.bfloat16, device="cuda:0",
                        load_encoders: bool = True, disable_guardrails: bool = True):
        """``ChronoEditPipeline.from_pretrained(model_path, image_encoder=..., transformer=..., vae=..., torch_dtype=bf16,
        disable_guardrails=...)`` (run_inference_diffusers.py:357-364): components handed in are used as they are, the others
        are read from the diffusers directory layout (``transformer/``, ``vae/``, ``text_encoder/``, ``image_encoder/``,
        ``scheduler/scheduler_config.json``; ``tokenizer/`` and ``image_processor/`` through transformers when present).
        The scheduler defaults to the diffusers sigma grid - the class the reference runner installs (:379-382)."""
        import json
        import os

        from .vae import AutoencoderKLWan
        if transformer is None:
            transformer = ChronoEditTransformer3DModel.from_pretrained(path, subfolder="transformer", torch_dtype=torch_dtype, device=device)
        if vae is None:
            vae = AutoencoderKLWan.from_pretrained(path, subfolder="vae", torch_dtype=torch_dtype, device=device)
        if scheduler is None:
            cfg_file = os.path.join(path, "scheduler", "scheduler_config.json")
            cfg = {}
            if os.path.exists(cfg_file):
                with open(cfg_file) as f:
                    cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
            cfg.setdefault("sigma_grid", "diffusers")
            scheduler = FlowUniPCMultistepScheduler.from_config(cfg)
        if load_encoders:
            if text_encoder is None and os.path.isdir(os.path.join(path, "text_encoder")):
                from .umt5 import UMT5EncoderModel
                text_encoder = UMT5EncoderModel.from_pretrained(path, subfolder="text_encoder", torch_dtype=torch_dtype, device=device)
            if image_encoder is None and os.path.isdir(os.path.join(path, "image_encoder")):
                from .clip_vision import CLIPVisionModel
                image_encoder = CLIPVisionModel.from_pretrained(path, subfolder="image_encoder", torch_dtype=torch_dtype, device=device)
            if tokenizer is None and os.path.isdir(os.path.join(path, "tokenizer")):
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
            if image_processor is None and os.path.isdir(os.path.join(path, "image_processor")):
                from transformers import CLIPImageProcessor
                image_processor = CLIPImageProcessor.from_pretrained(os.path.join(path, "image_processor"))
        return cls(tokenizer=tokenizer, text_encoder=text_encoder, image_encoder=image_encoder, image_processor=image_processor,
                   transformer=transformer, vae=vae, scheduler=scheduler, disable_guardrails=disable_guardrails)

    # -- conditioning (pipeline_chronoedit.py:205-330) -------------------------------------------------------------------
    def _get_t5_prompt_embeds(self, prompt=None, num_videos_per_prompt: int = 1, max_sequence_length: int = 512, device=None, dtype=None):
        """:205-243: clean, tokenise to max_length, encode, zero the padding rows, repeat per video."""
        if self.tokenizer is None or self.text_encoder is None:
            raise ValueError("`prompt` strings need the pipeline's tokenizer and text_encoder; pass `prompt_embeds` instead")
        from .umt5 import t5_prompt_embeds
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        prompt = [prompt_clean(u) for u in prompt]
        batch_size = len(prompt)
        text_inputs = self.tokenizer(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                                     add_special_tokens=True, return_attention_mask=True, return_tensors="pt")
        ids, mask = text_inputs.input_ids, text_inputs.attention_mask
        prompt_embeds = t5_prompt_embeds(self.text_encoder, ids.to(device), mask.to(device))
        if dtype is not None:
            prompt_embeds = prompt_embeds.to(dtype)
        _, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_videos_per_prompt, 1)
        return prompt_embeds.view(batch_size * num_videos_per_prompt, seq_len, -1)

    def encode_prompt(self, prompt=None, negative_prompt=None, do_classifier_free_guidance: bool = True, num_videos_per_prompt: int = 1,
                      prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                      max_sequence_length: int = 226, device=None, dtype=None, *, input_ids: Optional[torch.Tensor] = None,
                      attention_mask: Optional[torch.Tensor] = None, negative_input_ids: Optional[torch.Tensor] = None,
                      negative_attention_mask: Optional[torch.Tensor] = None):
        """:258-330 (strings through the injected tokenizer).  Tokenizer-free form for callers that tokenise themselves:
        `encode_prompt(input_ids=, attention_mask=, negative_input_ids=, negative_attention_mask=)` (ids padded to max length)."""
        if input_ids is not None:
            from .umt5 import t5_prompt_embeds
            if self.text_encoder is None:
                raise ValueError("this pipeline was built without a text_encoder: pass prompt_embeds instead")
            pos = t5_prompt_embeds(self.text_encoder, input_ids, attention_mask)
            neg = None if negative_input_ids is None else t5_prompt_embeds(self.text_encoder, negative_input_ids, negative_attention_mask)
            return pos, neg
        prompt = [prompt] if isinstance(prompt, str) else prompt
        batch_size = len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds = self._get_t5_prompt_embeds(prompt, num_videos_per_prompt, max_sequence_length, device, dtype)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt = batch_size * [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: {prompt} has "
                                 f"batch size {batch_size}. Please make sure that passed `negative_prompt` matches the batch size of `prompt`.")
            negative_prompt_embeds = self._get_t5_prompt_embeds(negative_prompt, num_videos_per_prompt, max_sequence_length, device, dtype)
        return prompt_embeds, negative_prompt_embeds

    def encode_image(self, image, device=None) -> torch.Tensor:
        """:247-256: CLIP penultimate hidden state.  `image`: whatever the injected CLIPImageProcessor accepts, or - without a
        processor - the pixel_values tensor [B,3,224,224] it would have produced."""
        if self.image_encoder is None:
            raise ValueError("this pipeline was built without an image_encoder: pass image_embeds instead")
        device = device or self._execution_device
        if self.image_processor is not None and not (isinstance(image, torch.Tensor) and image.dim() == 4 and image.is_floating_point()
                                                     and image.shape[1] == 3 and image.min() < 0):
            pixel_values = self.image_processor(images=image, return_tensors="pt")["pixel_values"]
        elif isinstance(image, torch.Tensor):
            pixel_values = image
        else:
            raise ValueError("encode_image needs the pipeline's image_processor for non-tensor images")
        return self.image_encoder(pixel_values=pixel_values.to(device), output_hidden_states=True).hidden_states[-2]

    def check_inputs(self, prompt, negative_prompt, image, height, width, prompt_embeds=None, negative_prompt_embeds=None,
                     image_embeds=None, callback_on_step_end_tensor_inputs=None):
        """:332-390, message for message - with ONE rule relaxed: the reference rejects `image` together with `image_embeds`
        although its own `__call__` needs `image` for `prepare_latents`, which makes `image_embeds` unusable there; here
        both may be given (the embeds then skip the CLIP encoder)."""
        if image is None and image_embeds is None:
            raise ValueError("Provide either `image` or `prompt_embeds`. Cannot leave both `image` and `image_embeds` undefined.")
        if image is not None and not isinstance(image, torch.Tensor) and not _is_pil(image):
            raise ValueError(f"`image` has to be of type `torch.Tensor` or `PIL.Image.Image` but is {type(image)}")
        if height % 16 != 0 or width % 16 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 16 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None and not all(k in self._callback_tensor_inputs for k in callback_on_step_end_tensor_inputs):
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found "
                             f"{[k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to only forward one of the two.")
        elif negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `negative_prompt`: {negative_prompt} and `negative_prompt_embeds`: {negative_prompt_embeds}. "
                             "Please make sure to only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        elif negative_prompt is not None and (not isinstance(negative_prompt, str) and not isinstance(negative_prompt, list)):
            raise ValueError(f"`negative_prompt` has to be of type `str` or `list` but is {type(negative_prompt)}")

    def prepare_latents(self, image: torch.Tensor, batch_size: int, num_channels_latents: int = 16, height: int = 480, width: int = 832,
                        num_frames: int = 81, dtype: Optional[torch.dtype] = None, device=None, generator=None,
                        latents: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """:392-456 with the reference's argument list: returns (latents in `dtype`, condition [B,20,T,h,w])."""
        device = device or self._execution_device
        T = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        shape = (batch_size, num_channels_latents, T, height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = _randn_tensor(shape, generator, device, dtype or torch.bfloat16)
        else:
            latents = latents.to(device=device, dtype=dtype or latents.dtype)
        if image.shape[-2:] != (height, width):
            raise ValueError(f"image is {tuple(image.shape[-2:])}, expected ({height}, {width}) after preprocessing")
        _, condition = prepare_latents(self.vae, image.to(device), num_frames, latents=latents)
        if condition.shape[0] != batch_size:
            condition = condition.repeat(batch_size, 1, 1, 1, 1)
        return latents, condition

    # LoRA entry points of the reference runner (run_inference_diffusers.py:370-374); the adapters target the transformer
    def load_lora_weights(self, path_or_state, adapter_name: str = "default"):
        self.transformer.load_lora_weights(path_or_state, adapter_name=adapter_name)
        return self

    def fuse_lora(self, adapter_names=None, lora_scale: float = 1.0):
        self.transformer.fuse_lora(adapter_names=adapter_names, lora_scale=lora_scale)
        return self

    # -- image pre / post processing (diffusers VideoProcessor, pipeline_chronoedit.py:673,801) ---------------------------
    @staticmethod
    def preprocess_image(image, height: int, width: int) -> torch.Tensor:
        """VideoProcessor.preprocess(image, height=, width=): PIL / array / tensor -> [B,3,height,width] in [-1, 1] (fp32, CPU or
        the tensor's device).  PIL images are resized with Lanczos, arrays / tensors with F.interpolate's default mode (nearest) - both as
        diffusers' VaeImageProcessor.resize does; arrays / tensors are taken as [0, 1] unless they already carry negative values
        (diffusers' own rule)."""
        import numpy as np
        if _is_pil(image) or (isinstance(image, (list, tuple)) and all(_is_pil(i) for i in image)):
            from PIL import Image
            imgs = list(image) if isinstance(image, (list, tuple)) else [image]
            arr = np.stack([np.asarray(i.convert("RGB").resize((width, height), Image.LANCZOS), dtype=np.float32) / 255.0 for i in imgs])
            x = torch.from_numpy(arr).permute(0, 3, 1, 2)
        elif isinstance(image, np.ndarray):
            x = torch.from_numpy(image.astype(np.float32))
            x = x[None] if x.dim() == 3 else x
            x = x.permute(0, 3, 1, 2)
        elif isinstance(image, torch.Tensor):
            x = image.float()
            x = x[None] if x.dim() == 3 else x
        else:
            raise ValueError(f"`image` has to be of type `torch.Tensor` or `PIL.Image.Image` but is {type(image)}")
        if x.shape[-2:] != (height, width):
            x = torch.nn.functional.interpolate(x, size=(height, width))  # arrays / tensors: F.interpolate's default (nearest), as diffusers' VaeImageProcessor.resize
        if x.min() >= 0:
            x = 2.0 * x - 1.0
        return x.contiguous()

    @staticmethod
    def postprocess_video(video: torch.Tensor, output_type: str = "np"):
        """VideoProcessor.postprocess_video: [B,3,F,H,W] in [-1,1] -> per sample F frames in [0,1]: "np" [B,F,H,W,3] float32,
        "pt" [B,F,3,H,W], "pil" list of lists of PIL images."""
        v = (video.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 1, 3, 4)  # [B,F,3,H,W]
        if output_type == "pt":
            return v
        arr = v.permute(0, 1, 3, 4, 2).cpu().numpy()
        if output_type == "np":
            return arr
        if output_type == "pil":
            from PIL import Image
            return [[Image.fromarray((f * 255).round().astype("uint8")) for f in sample] for sample in arr]
        raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil']")

    # -- the call (pipeline_chronoedit.py:484-812) -----------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, image=None, prompt: Union[str, List[str], None] = None, negative_prompt: Union[str, List[str], None] = None,
                 height: int = 480, width: int = 832, num_frames: int = 81, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 num_videos_per_prompt: Optional[int] = 1, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 image_embeds: Optional[torch.Tensor] = None, output_type: Optional[str] = "np", return_dict: bool = True,
                 attention_kwargs: Optional[Dict[str, Any]] = None, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 512,
                 enable_temporal_reasoning: bool = False, num_temporal_reasoning_steps: int = 0, offload_model: bool = False):
        """Same arguments, defaults, checks and return value as the reference's `__call__`.  What differs is underneath: the two
        guidance passes run as ONE batched forward, CFG + flow-UniPC are one fused launch, the latents stay fp32 on the device
        (`scheduler.trajectory_dtype = torch.bfloat16` restores the reference's bf16 rounding of latents and history)."""
        if hasattr(callback_on_step_end, "tensor_inputs"):
            callback_on_step_end_tensor_inputs = callback_on_step_end.tensor_inputs
        self.check_inputs(prompt, negative_prompt, image, height, width, prompt_embeds, negative_prompt_embeds, image_embeds,
                          callback_on_step_end_tensor_inputs)
        if num_frames % self.vae_scale_factor_temporal != 1:  # :606-611
            num_frames = num_frames // self.vae_scale_factor_temporal * self.vae_scale_factor_temporal + 1
        num_frames = max(num_frames, 1)
        self._guidance_scale, self._attention_kwargs, self._current_timestep, self._interrupt = guidance_scale, attention_kwargs, None, False
        device = self._execution_device
        if self.text_guardrail_runner is not None and not self.text_guardrail_runner(prompt):  # :621-629
            raise Exception(f"Guardrail blocked text2world generation. Prompt: {prompt}")

        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt=prompt, negative_prompt=negative_prompt, do_classifier_free_guidance=self.do_classifier_free_guidance,
            num_videos_per_prompt=num_videos_per_prompt, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
            max_sequence_length=max_sequence_length, device=device)
        if offload_model and self.text_encoder is not None:
            self.text_encoder.cpu()
        tdtype = self.transformer.dtype
        prompt_embeds = prompt_embeds.to(device=device, dtype=tdtype)
        if negative_prompt_embeds is not None:
            negative_prompt_embeds = negative_prompt_embeds.to(device=device, dtype=tdtype)
        if image_embeds is None:
            image_embeds = self.encode_image(image, device)
        image_embeds = image_embeds.to(device=device, dtype=tdtype)
        if image_embeds.shape[0] != batch_size:
            image_embeds = image_embeds.repeat(batch_size, 1, 1)
        if offload_model and self.image_encoder is not None:
            self.image_encoder.cpu()

        B = batch_size * num_videos_per_prompt
        if hasattr(self.vae, "use_graph"):  # the VAE replays captured graphs from the second edit of a shape on (vae.py)
            self.vae.use_graph = bool(self.use_graph)
        img = self.preprocess_image(image, height, width).to(device=device, dtype=torch.bfloat16)
        latents, condition = self.prepare_latents(img, B, self.vae.config.z_dim, height, width, num_frames, torch.bfloat16, device,
                                                  generator, latents)
        if offload_model and hasattr(self.vae, "clear_graphs"):
            self.vae.clear_graphs()  # a captured encode graph pins its activation pool next to the 14B DiT: the low-memory mode drops it
        if prompt_embeds.shape[0] != B or (negative_prompt_embeds is not None and negative_prompt_embeds.shape[0] != B):
            raise ValueError(f"prompt_embeds carry {prompt_embeds.shape[0]} samples, expected batch_size * num_videos_per_prompt = {B}")
        if image_embeds.shape[0] != B:
            image_embeds = image_embeds.repeat(B // image_embeds.shape[0], 1, 1)

        # The engine denoises ONE edit at a time (its batch axis carries the guidance pair); B = batch_size * num_videos_per_prompt
        # edits (pipeline_chronoedit.py:493,631-637,676-691) run one after the other on the same weights - the reference's batched
        # loop computes the same per-sample arithmetic.  `callback_on_step_end` is then called per sample and step with that
        # sample's tensors; returned `latents`, `prompt_embeds`, `negative_prompt_embeds` are honoured (:741-749).
        self._num_timesteps = num_inference_steps
        done = []
        for b in range(B):
            embeds = {"prompt_embeds": prompt_embeds[b:b + 1], "negative_prompt_embeds": None if negative_prompt_embeds is None else negative_prompt_embeds[b:b + 1]}

            def on_step_end(i, t, lat, embeds=embeds):
                self._current_timestep = t
                if callback_on_step_end is None:
                    return None
                pool = {"latents": lat, **embeds}
                outs = callback_on_step_end(self, i, t, {k: pool[k] for k in callback_on_step_end_tensor_inputs})
                if not outs:
                    return None
                new = {k: outs[k] for k in ("latents", "prompt_embeds", "negative_prompt_embeds") if k in outs and outs[k] is not pool[k]}
                embeds.update({k: v for k, v in new.items() if k != "latents"})
                return new or None

            done.append(denoise(self.transformer, self.scheduler, latents[b:b + 1], condition[b:b + 1], embeds["prompt_embeds"],
                                embeds["negative_prompt_embeds"] if self.do_classifier_free_guidance else None, image_embeds[b:b + 1],
                                num_inference_steps, guidance_scale, enable_temporal_reasoning, num_temporal_reasoning_steps,
                                use_graph=self.use_graph, on_step_end=on_step_end, interrupted=lambda: self._interrupt,
                                graph_warm=self._graph_warm))
        latents = done[0] if B == 1 else torch.cat(done, dim=0)
        if offload_model and self.transformer is not None:
            self.transformer.cpu()
            torch.cuda.empty_cache()
        self._current_timestep = None

        if output_type != "latent":
            video = decode_latents(self.vae, latents, enable_temporal_reasoning, num_temporal_reasoning_steps)
            if self.video_guardrail_runner is not None:  # :784-799
                video = self.video_guardrail_runner(video)
                if video is None:
                    raise Exception("Guardrail blocked video2world generation.")
            video = self.postprocess_video(video, output_type=output_type)
        else:
            video = latents
        if offload_model and hasattr(self.vae, "clear_graphs"):
            self.vae.clear_graphs()
            torch.cuda.empty_cache()
        self.maybe_free_model_hooks()
        if not return_dict:
            return (video,)
        return WanPipelineOutput(frames=video)

    @torch.no_grad()
    def edit_tensors(self, image: torch.Tensor, prompt_embeds: torch.Tensor, negative_prompt_embeds: Optional[torch.Tensor],
                     image_embeds: Optional[torch.Tensor], num_frames: int = 5, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                     enable_temporal_reasoning: bool = False, num_temporal_reasoning_steps: int = 0, generator=None,
                     latents: Optional[torch.Tensor] = None, output_type: str = "pt"):
        """Tensor-level form of the same edit (no pre / post processing): image [1,3,H,W] in [-1,1] -> video [1,3,F,H,W] in
        [-1,1] ("pt") or the final latents ("latent").  What the parity tests and tools/full_edit.py drive."""
        H, W = image.shape[-2:]
        if H % 16 != 0 or W % 16 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 16 but are {H} and {W}.")  # pipeline_chronoedit.py:361-362
        if num_frames % 4 != 1:
            num_frames = max(num_frames // 4 * 4 + 1, 1)  # :606-611
        latents, condition = prepare_latents(self.vae, image, num_frames, latents, generator)
        latents = denoise(self.transformer, self.scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
                          num_inference_steps, guidance_scale, enable_temporal_reasoning, num_temporal_reasoning_steps,
                          use_graph=self.use_graph, graph_warm=self._graph_warm)
        if output_type == "latent":
            return latents
        return decode_latents(self.vae, latents, enable_temporal_reasoning, num_temporal_reasoning_steps)
