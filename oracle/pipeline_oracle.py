"""CPU oracle for one whole edit (TEST INFRASTRUCTURE): prepare_latents -> N-step CFG loop -> decode, restating
/root/reference/chronoedit_diffusers/pipeline_chronoedit.py:392-456 and :694-781 on top of the DiT, UniPC and VAE oracles.
Computation dtype is the caller's (fp32 = BASELINE.json configs[0] "CPU float32 eager", bf16 = the reference's run mode)."""
from __future__ import annotations

import torch

from . import dit_oracle as D
from . import vae_oracle as V
from .unipc_oracle import UniPCOracle


def prepare_latents(vp, vcfg, image, num_frames, latents):
    B, _, H, W = image.shape
    tds = 2 ** sum(vcfg.temperal_downsample)
    h, w = H // 8, W // 8
    video = torch.cat([image.unsqueeze(2), image.new_zeros(B, 3, num_frames - 1, H, W)], dim=2)
    mean = torch.tensor(V.LATENTS_MEAN[: vcfg.z_dim], dtype=image.dtype).view(1, -1, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(V.LATENTS_STD[: vcfg.z_dim])).to(image.dtype).view(1, -1, 1, 1, 1)
    cond = (V.encode(vp, vcfg, video) - mean) * inv_std
    mask = torch.ones(B, 1, num_frames, h, w)
    mask[:, :, 1:] = 0
    first = torch.repeat_interleave(mask[:, :, 0:1], dim=2, repeats=tds)
    mask = torch.cat([first, mask[:, :, 1:]], dim=2).view(B, -1, tds, h, w).transpose(1, 2)
    return latents, torch.cat([mask.to(cond.dtype), cond], dim=1)


def edit(dp, dcfg, vp, vcfg, image, prompt, negative, image_embeds, latents, num_frames=5, steps=4, guidance=5.0, shift=5.0,
         decode=True, enable_temporal_reasoning=False, num_temporal_reasoning_steps=0, fp8=False):
    latents, cond = prepare_latents(vp, vcfg, image, num_frames, latents)
    sch = UniPCOracle()
    sch.set_timesteps(steps, shift=shift)
    for i, t in enumerate(sch.timesteps):
        if enable_temporal_reasoning and i == num_temporal_reasoning_steps:  # pipeline_chronoedit.py:700-709
            latents = latents[:, :, [0, -1]]
            cond = cond[:, :, [0, -1]]
            sch.model_outputs = [None if m is None else (m[:, :, [0, -1]] if m.shape[-3] != latents.shape[-3] else m)
                                 for m in sch.model_outputs]
            if sch.last_sample is not None:
                sch.last_sample = sch.last_sample[:, :, [0, -1]] if sch.last_sample.shape[-3] != latents.shape[-3] else sch.last_sample
        inp = torch.cat([latents, cond], dim=1).to(prompt.dtype)
        ts = t.expand(latents.shape[0])
        c = D.dit_forward(dp, dcfg, inp, ts, prompt, image_embeds, fp8=fp8)
        if guidance > 1.0:
            u = D.dit_forward(dp, dcfg, inp, ts, negative, image_embeds, fp8=fp8)
            c = u + guidance * (c - u)
        latents = sch.step(c.to(latents.dtype), latents)
    if not decode:
        return latents, None
    mean = torch.tensor(V.LATENTS_MEAN[: vcfg.z_dim], dtype=latents.dtype).view(1, -1, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(V.LATENTS_STD[: vcfg.z_dim])).to(latents.dtype).view(1, -1, 1, 1, 1)
    z = latents / inv_std + mean
    if enable_temporal_reasoning and num_temporal_reasoning_steps > 0:  # pipeline_chronoedit.py:776-779
        video_edit = V.decode(vp, vcfg, z[:, :, [0, -1]])
        video_reason = V.decode(vp, vcfg, z[:, :, :-1])
        video = torch.cat([video_reason, video_edit[:, :, 1:]], dim=2)
    else:
        video = V.decode(vp, vcfg, z)
    return latents, video
