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


@torch.no_grad()
def denoise(transformer, scheduler, latents, condition, prompt_embeds, negative_prompt_embeds, image_embeds,
            num_inference_steps: int, guidance_scale: float = 5.0, enable_temporal_reasoning: bool = False,
            num_temporal_reasoning_steps: int = 0):
    """The whole loop, including the temporal-reasoning truncation 8 -> 2 latent frames (pipeline_chronoedit.py:700-709)."""
    scheduler.set_timesteps(num_inference_steps, device=latents.device)
    latents = latents.to(torch.float32).contiguous()
    cfg_inputs = None
    if guidance_scale > 1.0 and negative_prompt_embeds is not None:
        cfg_inputs = make_cfg_inputs(prompt_embeds, negative_prompt_embeds, image_embeds)
    for i, t in enumerate(scheduler.timesteps):
        if enable_temporal_reasoning and i == num_temporal_reasoning_steps:
            latents = latents[:, :, [0, -1]].contiguous()
            condition = condition[:, :, [0, -1]].contiguous()
            for j in range(len(scheduler.model_outputs)):
                mo = scheduler.model_outputs[j]
                if mo is not None and mo.shape[-3] != latents.shape[-3]:
                    scheduler.model_outputs[j] = mo[:, :, [0, -1]].contiguous()
            if scheduler.last_sample is not None and scheduler.last_sample.shape[-3] != latents.shape[-3]:
                scheduler.last_sample = scheduler.last_sample[:, :, [0, -1]].contiguous()
        latents = denoise_step(transformer, scheduler, latents, condition, t, prompt_embeds, negative_prompt_embeds,
                               image_embeds, guidance_scale, cfg_inputs=cfg_inputs)
    return latents
