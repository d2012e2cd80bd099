"""Denoising loop of ChronoEditPipeline.__call__ on the HIP engine
(/root/reference/chronoedit_diffusers/pipeline_chronoedit.py:694-756).

`denoise_step` is one loop iteration — the unit BASELINE.json's "denoising-steps/sec" counts:
cat(latents, condition) -> 1 or 2 DiT forwards -> CFG -> scheduler update.  The latents stay fp32
on the device for the whole trajectory; the CFG combine and the UniPC update are one fused launch.
"""
from __future__ import annotations

from typing import Optional

import torch

from .scheduler import FlowUniPCMultistepScheduler
from .transformer import ChronoEditTransformer3DModel


def make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds):
    """[cond | uncond] conditioning stacked along the batch axis; build it ONCE per edit so that the transformer's
    context cache (step-invariant text/image K/V) can recognise the tensors across steps."""
    text2 = torch.cat([prompt_embeds, negative_prompt_embeds], 0)
    image2 = None if image_embeds is None else torch.cat([image_embeds, image_embeds], 0)
    return text2, image2


@torch.no_grad()
def denoise_step(transformer: ChronoEditTransformer3DModel, scheduler: FlowUniPCMultistepScheduler, latents: torch.Tensor,
                 condition: torch.Tensor, t: torch.Tensor, prompt_embeds: torch.Tensor,
                 negative_prompt_embeds: Optional[torch.Tensor], image_embeds: Optional[torch.Tensor],
                 guidance_scale: float, batch_cfg: bool = True, cfg_inputs=None) -> torch.Tensor:
    """latents fp32 [B,16,T,h,w] (updated in place), condition bf16 [B,20,T,h,w]  (pipeline_chronoedit.py:711-739)."""
    latent_model_input = torch.cat([latents.to(torch.bfloat16), condition], dim=1)
    B = latents.shape[0]
    timestep = t.expand(B)
    if guidance_scale > 1.0 and negative_prompt_embeds is not None:  # do_classifier_free_guidance
        if batch_cfg:
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


class GraphedDenoiser:
    """One denoising step captured as a hipGraph and replayed per step (north star: "the 50-step / 8-step sampling loops
    are hipGraph-captured").  Everything a step reads lives at fixed device addresses: the fp32 latents (updated in place),
    the condition, the stacked CFG conditioning, the scheduler history, and two small staging buffers that receive the
    step's timestep and UniPC coefficient row (device-to-device copies enqueued in front of the replay - no host sync).
    A new graph is needed when the latent shape changes (temporal-reasoning truncation 8 -> 2 frames)."""

    def __init__(self, transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
                 guidance_scale: float, batch_cfg: bool = True):
        assert latents.dtype == torch.float32 and latents.is_contiguous()
        self.tr, self.sch, self.latents, self.condition = transformer, scheduler, latents, condition
        self.prompt, self.negative, self.image, self.g, self.batch_cfg = prompt_embeds, negative_prompt_embeds, image_embeds, guidance_scale, batch_cfg
        dev = latents.device
        self.cfg_inputs = None
        if guidance_scale > 1.0 and negative_prompt_embeds is not None:
            self.cfg_inputs = make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
        self.t_buf = torch.zeros((), dtype=torch.int64, device=dev)
        self.coef_buf = torch.zeros(10, dtype=torch.float32, device=dev)
        scheduler._ensure_state(latents)
        # warm-up (lazy initialisations: packed weights, workspaces, function attributes) on saved state, then capture
        saved = (latents.clone(), [m.clone() for m in scheduler.model_outputs], scheduler.last_sample.clone(), scheduler._step_index)
        self._stage(scheduler._step_index or 0)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._body()
        torch.cuda.current_stream().wait_stream(side)
        self._restore(saved)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        self._restore(saved)

    def _restore(self, saved):
        lat, mos, last, idx = saved
        self.latents.copy_(lat)
        for m, s in zip(self.sch.model_outputs, mos):
            m.copy_(s)
        self.sch.last_sample.copy_(last)
        self.sch._step_index = idx

    def _stage(self, i):
        self.t_buf.copy_(self.sch.timesteps[i])
        self.coef_buf.copy_(self.sch.coef_row(i, self.g, self.latents.device))

    def _body(self):
        inp = torch.cat([self.latents.to(torch.bfloat16), self.condition], dim=1)
        B = inp.shape[0]
        ts = self.t_buf.expand(B)
        if self.cfg_inputs is not None and self.batch_cfg:
            out = self.tr(torch.cat([inp, inp], 0), torch.cat([ts, ts], 0), self.cfg_inputs[0], self.cfg_inputs[1], return_dict=False)[0]
            c, u = out[:B].contiguous(), out[B:].contiguous()
        elif self.cfg_inputs is not None:
            c = self.tr(inp, ts, self.prompt, self.image, return_dict=False)[0]
            u = self.tr(inp, ts, self.negative, self.image, return_dict=False)[0]
        else:
            c, u = self.tr(inp, ts, self.prompt, self.image, return_dict=False)[0], None
        self.sch.step_cfg(c, u, self.g, self.latents, coef=self.coef_buf)

    def step(self, i: int) -> torch.Tensor:
        self._stage(i)
        self.graph.replay()
        self.sch._step_index = i + 1
        return self.latents


@torch.no_grad()
def denoise(transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
            num_inference_steps: int, guidance_scale: float = 5.0, enable_temporal_reasoning: bool = False,
            num_temporal_reasoning_steps: int = 0, use_graph: bool = False):
    """The whole loop, including the temporal-reasoning truncation 8 -> 2 latent frames (pipeline_chronoedit.py:700-709)."""
    scheduler.set_timesteps(num_inference_steps, device=latents.device)
    latents = latents.to(torch.float32).contiguous()
    cfg_inputs = None
    if guidance_scale > 1.0 and negative_prompt_embeds is not None:
        cfg_inputs = make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
    graphed = None
    for i, t in enumerate(scheduler.timesteps):
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
                graphed = GraphedDenoiser(transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds,
                                          image_embeds, guidance_scale)
            latents = graphed.step(i)
        else:
            latents = denoise_step(transformer, scheduler, latents, condition, t, prompt_embeds, negative_prompt_embeds,
                                   image_embeds, guidance_scale, cfg_inputs=cfg_inputs)
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


class ChronoEditPipeline:
    """The denoising part of the reference pipeline behind the same call (text / CLIP encoders and guardrails are out of
    scope: pass `prompt_embeds`, `negative_prompt_embeds`, `image_embeds` as the reference's encoders produce them)."""

    def __init__(self, vae, transformer: ChronoEditTransformer3DModel, scheduler: FlowUniPCMultistepScheduler,
                 text_encoder=None, image_encoder=None):
        self.vae, self.transformer, self.scheduler = vae, transformer, scheduler
        self.text_encoder, self.image_encoder = text_encoder, image_encoder  # chronoedit_amd.umt5 / .clip_vision drop-ins

    @classmethod
    def from_pretrained(cls, path: str, transformer=None, vae=None, text_encoder=None, image_encoder=None, scheduler=None,
                        torch_dtype: torch.dtype = torch.bfloat16, device="cuda:0", load_encoders: bool = True, **unused):
        """``ChronoEditPipeline.from_pretrained(model_path, image_encoder=..., transformer=..., vae=..., torch_dtype=bf16)``
        (run_inference_diffusers.py:357-364): components handed in are used as they are, the others are read from the
        diffusers directory layout (``transformer/``, ``vae/``, ``text_encoder/``, ``image_encoder/``,
        ``scheduler/scheduler_config.json``).  The tokenizer and the CLIP image processor are host-side transformers objects and
        stay with the caller."""
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
            scheduler = FlowUniPCMultistepScheduler.from_config(cfg)
        if load_encoders:
            if text_encoder is None and os.path.isdir(os.path.join(path, "text_encoder")):
                from .umt5 import UMT5EncoderModel
                text_encoder = UMT5EncoderModel.from_pretrained(path, subfolder="text_encoder", torch_dtype=torch_dtype, device=device)
            if image_encoder is None and os.path.isdir(os.path.join(path, "image_encoder")):
                from .clip_vision import CLIPVisionModel
                image_encoder = CLIPVisionModel.from_pretrained(path, subfolder="image_encoder", torch_dtype=torch_dtype, device=device)
        return cls(vae, transformer, scheduler, text_encoder=text_encoder, image_encoder=image_encoder)

    def encode_prompt(self, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                      negative_input_ids: Optional[torch.Tensor] = None, negative_attention_mask: Optional[torch.Tensor] = None):
        """``encode_prompt`` / ``_get_t5_prompt_embeds`` after the tokenizer (pipeline_chronoedit.py:205-243,258-330): token ids
        padded to ``max_sequence_length`` and their masks in, ``(prompt_embeds, negative_prompt_embeds)`` out.  Tokenising
        (``prompt_clean`` + the UMT5 sentencepiece tokenizer) stays the reference's host code."""
        from .umt5 import t5_prompt_embeds
        if self.text_encoder is None:
            raise ValueError("this pipeline was built without a text_encoder: pass prompt_embeds instead")
        pos = t5_prompt_embeds(self.text_encoder, input_ids, attention_mask)
        neg = None if negative_input_ids is None else t5_prompt_embeds(self.text_encoder, negative_input_ids, negative_attention_mask)
        return pos, neg

    def encode_image(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """``encode_image`` after the CLIP image processor (pipeline_chronoedit.py:247-256): penultimate hidden state."""
        if self.image_encoder is None:
            raise ValueError("this pipeline was built without an image_encoder: pass image_embeds instead")
        return self.image_encoder(pixel_values=pixel_values, output_hidden_states=True).hidden_states[-2]

    # LoRA entry points of the reference runner (run_inference_diffusers.py:370-374); the adapters target the transformer
    def load_lora_weights(self, path_or_state, adapter_name: str = "default"):
        self.transformer.load_lora_weights(path_or_state, adapter_name=adapter_name)
        return self

    def fuse_lora(self, adapter_names=None, lora_scale: float = 1.0):
        self.transformer.fuse_lora(adapter_names=adapter_names, lora_scale=lora_scale)
        return self

    @torch.no_grad()
    def __call__(self, image: torch.Tensor, prompt_embeds: torch.Tensor, negative_prompt_embeds: Optional[torch.Tensor],
                 image_embeds: Optional[torch.Tensor], num_frames: int = 5, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 enable_temporal_reasoning: bool = False, num_temporal_reasoning_steps: int = 0, generator=None,
                 latents: Optional[torch.Tensor] = None, output_type: str = "pt"):
        H, W = image.shape[-2:]
        if H % 16 != 0 or W % 16 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 16 but are {H} and {W}.")  # pipeline_chronoedit.py:361-362
        if num_frames % 4 != 1:
            num_frames = max(num_frames // 4 * 4 + 1, 1)  # :606-611
        latents, condition = prepare_latents(self.vae, image, num_frames, latents, generator)
        latents = denoise(self.transformer, self.scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
                          num_inference_steps, guidance_scale, enable_temporal_reasoning, num_temporal_reasoning_steps)
        if output_type == "latent":
            return latents
        return decode_latents(self.vae, latents, enable_temporal_reasoning, num_temporal_reasoning_steps)
