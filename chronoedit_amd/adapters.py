"""Fronts for the reference's two SIBLING stacks over the same HIP engine.

The reference ships the ChronoEdit DiT three times: the diffusers class this package mirrors in ``transformer.py``, the
DiffSynth ``WanModel`` driven by ``model_fn_wan_video`` (chronoedit_diffsynth/wan_video_dit_chronoedit.py:287-431,
wan_video_new_chronoedit.py:1296-1504) and the imaginaire ``EditWanModel`` (chronoedit/_src/networks/wan2pt1.py:600-860,
chronoedit_14b.py:137-162).  The two siblings keep the original Wan parameter names and differ from the diffusers front in
their call signatures only:

=====================  ===============================================  ============================================
                       DiffSynth                                        imaginaire
=====================  ===============================================  ============================================
latents / condition    ``latents`` (``x``) [B,16,T,h,w] and ``y``        ``x_B_C_T_H_W`` and ``y_B_C_T_H_W``, concatenated
                       [B,20,T,h,w], concatenated inside (:1408-1409)   inside (wan2pt1.py:786-787)
timestep               float tensor [B] (or [1], broadcast :1403-1404)  ``timesteps_B_T`` [B,1] (:780-781)
text / CLIP tokens     ``context`` [B,L,4096], ``clip_feature``          ``crossattn_emb``, ``frame_cond_crossattn_emb_B_L_D``
                       [B,257,1280]; CLIP tokens first (:1410-1412)     (CLIP first, :826-828)
temporal RoPE          plain 0..T-1 in ``model_fn_wan_video`` (:1428);   {0, skip_len-1} for two latent frames
                       {0, skip_len-1} in ``WanModel.forward`` (:393)   (chronoedit_14b.py:112-131)
returns                tensor [B,16,T,h,w]                              tensor [B,16,T,h,w]
=====================  ===============================================  ============================================

Both classes own a :class:`~chronoedit_amd.transformer.ChronoEditTransformer3DModel`; ``load_state_dict`` /
``state_dict`` speak the native names (correspondence pinned by tests/golden/wan_native_keymap.json, semantics by
tests/golden/wan_native_tiny.pt, both produced by the reference's own files).  Options of the sibling call paths that the
ChronoEdit configurations never use (motion controller, VACE, audio, camera control, reference latents, TeaCache,
sliding windows, xfuser sequence parallelism, skip-layer guidance) raise ``NotImplementedError`` instead of being
ignored.  There is no CPU path: the HIP library must be present.
"""
from typing import Dict, Optional, Sequence

import torch
from torch import nn

from .transformer import ChronoEditTransformer3DModel

_CLIP_DIM = 1280  # MLP(1280, dim) in both siblings (wan_video_dit_chronoedit.py:348, wan2pt1.py:716)


def _build(dim, in_dim, ffn_dim, out_dim, text_dim, freq_dim, eps, patch_size, num_heads, num_layers, has_image, skip_len, device):
    if dim % num_heads:
        raise ValueError(f"dim {dim} is not a multiple of num_heads {num_heads}")
    return ChronoEditTransformer3DModel(
        patch_size=tuple(patch_size), num_attention_heads=num_heads, attention_head_dim=dim // num_heads, in_channels=in_dim,
        out_channels=out_dim, text_dim=text_dim, freq_dim=freq_dim, ffn_dim=ffn_dim, num_layers=num_layers, eps=eps,
        image_dim=_CLIP_DIM if has_image else None, added_kv_proj_dim=dim if has_image else None,
        rope_temporal_skip_len=skip_len, device=device)


def _reject(**options):
    used = [k for k, v in options.items() if v is not None and v is not False]
    if used:
        raise NotImplementedError(f"not part of the ChronoEdit hot path, not built: {', '.join(sorted(used))}")


class _NativeFront(nn.Module):
    """Shared plumbing: the engine, native-name (de)serialisation, one forward over (latents, condition, float timestep)."""

    transformer: ChronoEditTransformer3DModel

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):  # noqa: D401
        own = self.transformer.wan_native_state_dict()
        missing, unexpected = sorted(set(own) - set(state_dict)), sorted(set(state_dict) - set(own))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:4]} unexpected {unexpected[:4]}")
        self.transformer.load_wan_native_state_dict({k: v for k, v in state_dict.items() if k in own})
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def state_dict(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        return self.transformer.wan_native_state_dict()

    @property
    def dtype(self):
        return self.transformer.dtype

    def _run(self, latents, cond, timestep, text, clip, plain_temporal: bool):
        B = text.shape[0]
        if latents.shape[0] != B:  # "merged cfg": one latent, several prompts (wan_video_new_chronoedit.py:1399-1404)
            latents = torch.cat([latents] * B, dim=0)
        if cond is not None and cond.shape[0] != B:
            cond = torch.cat([cond] * (B // cond.shape[0]), dim=0)
        x = latents if cond is None else torch.cat([latents, cond], dim=1)
        t = timestep.reshape(-1)
        if not t.is_floating_point():
            t = t.to(torch.float32)
        if t.numel() != B:
            t = t.expand(B) if t.numel() == 1 else torch.cat([t] * (B // t.numel()))
        self.transformer.rope_plain_temporal = plain_temporal
        try:
            return self.transformer(x, t.to(torch.float32), text, clip, return_dict=False)[0]
        finally:
            self.transformer.rope_plain_temporal = False


class WanModel(_NativeFront):
    """DiffSynth front (wan_video_dit_chronoedit.py:287-431): same constructor keywords, native parameter names."""

    def __init__(self, dim: int, in_dim: int, ffn_dim: int, out_dim: int, text_dim: int, freq_dim: int, eps: float,
                 patch_size: Sequence[int], num_heads: int, num_layers: int, has_image_input: bool, has_image_pos_emb: bool = False,
                 has_ref_conv: bool = False, add_control_adapter: bool = False, in_dim_control_adapter: int = 24,
                 seperated_timestep: bool = False, require_vae_embedding: bool = True, require_clip_embedding: bool = True,
                 fuse_vae_embedding_in_latents: bool = False, rope_temporal_skip_len: int = 8, device=None):
        super().__init__()
        _reject(has_image_pos_emb=has_image_pos_emb, has_ref_conv=has_ref_conv, add_control_adapter=add_control_adapter,
                seperated_timestep=seperated_timestep, fuse_vae_embedding_in_latents=fuse_vae_embedding_in_latents)
        self.dim, self.in_dim, self.freq_dim, self.patch_size = dim, in_dim, freq_dim, tuple(patch_size)
        self.has_image_input = has_image_input
        self.require_vae_embedding, self.require_clip_embedding = require_vae_embedding, require_clip_embedding
        self.seperated_timestep, self.fuse_vae_embedding_in_latents = False, False
        self.rope_temporal_skip_len = rope_temporal_skip_len
        self.transformer = _build(dim, in_dim, ffn_dim, out_dim, text_dim, freq_dim, eps, patch_size, num_heads, num_layers,
                                  has_image_input, rope_temporal_skip_len, device)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, timestep: torch.Tensor, context: torch.Tensor, clip_feature: Optional[torch.Tensor] = None,
                y: Optional[torch.Tensor] = None, use_gradient_checkpointing: bool = False,
                use_gradient_checkpointing_offload: bool = False, **kwargs) -> torch.Tensor:
        """``WanModel.forward`` (:371-427): temporal positions {0, skip_len-1} for two latent frames (:393-397).  (The
        reference's method cannot run as shipped - it unpacks two values from ``patchify`` :391, which returns one
        :356-362 - this mirrors what it states.)"""
        _reject(use_gradient_checkpointing=use_gradient_checkpointing,
                use_gradient_checkpointing_offload=use_gradient_checkpointing_offload, **kwargs)
        if self.has_image_input and (y is None or clip_feature is None):
            raise ValueError("has_image_input: y (condition latents) and clip_feature are required")
        return self._run(x, y if self.has_image_input else None, timestep, context, clip_feature if self.has_image_input else None,
                         plain_temporal=False)


def model_fn_wan_video(dit: WanModel, motion_controller=None, vace=None, animate_adapter=None, latents: torch.Tensor = None,
                       timestep: torch.Tensor = None, context: torch.Tensor = None, clip_feature: Optional[torch.Tensor] = None,
                       y: Optional[torch.Tensor] = None, reference_latents=None, vace_context=None, vace_scale=1.0,
                       audio_embeds=None, motion_latents=None, s2v_pose_latents=None, drop_motion_frames: bool = True,
                       tea_cache=None, use_unified_sequence_parallel: bool = False, motion_bucket_id=None, pose_latents=None,
                       face_pixel_values=None, sliding_window_size=None, sliding_window_stride=None, cfg_merge: bool = False,
                       use_gradient_checkpointing: bool = False, use_gradient_checkpointing_offload: bool = False,
                       control_camera_latents_input=None, fuse_vae_embedding_in_latents: bool = False, **kwargs) -> torch.Tensor:
    """The function the DiffSynth pipeline runs per step (wan_video_new_chronoedit.py:95,1296-1504), same keywords, over a
    :class:`WanModel` of this module: PLAIN temporal positions (:1428-1432), latents broadcast to the prompt batch
    (:1399-1404).  ``cfg_merge`` only selects the sliding-window batch size in the reference (:1352) and is accepted."""
    _reject(motion_controller=motion_controller, vace=vace, animate_adapter=animate_adapter, reference_latents=reference_latents,
            vace_context=vace_context, audio_embeds=audio_embeds, motion_latents=motion_latents, s2v_pose_latents=s2v_pose_latents,
            tea_cache=tea_cache, use_unified_sequence_parallel=use_unified_sequence_parallel, motion_bucket_id=motion_bucket_id,
            pose_latents=pose_latents, face_pixel_values=face_pixel_values, sliding_window_size=sliding_window_size,
            sliding_window_stride=sliding_window_stride, use_gradient_checkpointing=use_gradient_checkpointing,
            use_gradient_checkpointing_offload=use_gradient_checkpointing_offload,
            control_camera_latents_input=control_camera_latents_input, fuse_vae_embedding_in_latents=fuse_vae_embedding_in_latents)
    if not isinstance(dit, WanModel):
        raise TypeError("dit must be a chronoedit_amd.adapters.WanModel")
    use_y = y is not None and dit.require_vae_embedding
    use_clip = clip_feature is not None and dit.require_clip_embedding
    if dit.has_image_input and not (use_y and use_clip):
        raise ValueError("this WanModel was built with has_image_input: y and clip_feature are required")
    return dit._run(latents, y if use_y else None, timestep, context, clip_feature if use_clip else None, plain_temporal=True)


class EditWanModel(_NativeFront):
    """imaginaire front (wan2pt1.py:600-860 + chronoedit_14b.py:137-162): same constructor keywords and forward signature."""

    def __init__(self, model_type: str = "i2v", patch_size: Sequence[int] = (1, 2, 2), text_len: int = 512, in_dim: int = 36,
                 dim: int = 5120, ffn_dim: int = 13824, freq_dim: int = 256, text_dim: int = 4096, out_dim: int = 16,
                 num_heads: int = 40, num_layers: int = 40, window_size=(-1, -1), qk_norm: bool = True, cross_attn_norm: bool = True,
                 eps: float = 1e-6, concat_padding_mask: bool = False, conv_patchify: bool = False, temporal_skip_p: bool = True,
                 temporal_skip_len: int = 10, device=None, **kwargs):
        super().__init__()
        if model_type not in ("t2v", "i2v", "flf2v"):
            raise ValueError(f"model_type {model_type!r}")
        _reject(concat_padding_mask=concat_padding_mask, window_attention=tuple(window_size) != (-1, -1),
                no_qk_norm=not qk_norm, no_cross_attn_norm=not cross_attn_norm, plain_video_rope=not temporal_skip_p)
        self.model_type, self.text_len, self.patch_size, self.dim = model_type, text_len, tuple(patch_size), dim
        self.temporal_skip_p, self.temporal_skip_len = temporal_skip_p, temporal_skip_len
        self.transformer = _build(dim, in_dim, ffn_dim, out_dim, text_dim, freq_dim, eps, patch_size, num_heads, num_layers,
                                  model_type in ("i2v", "flf2v"), temporal_skip_len, device)

    @torch.no_grad()
    def forward(self, x_B_C_T_H_W: torch.Tensor, timesteps_B_T: torch.Tensor, crossattn_emb: torch.Tensor, seq_len=None,
                frame_cond_crossattn_emb_B_L_D: Optional[torch.Tensor] = None, y_B_C_T_H_W: Optional[torch.Tensor] = None,
                padding_mask: Optional[torch.Tensor] = None, is_uncond: bool = False, slg_layers=None, **kwargs) -> torch.Tensor:
        """``WanModel.forward`` (wan2pt1.py:745-857) with the temporal-skip RoPE of ``EditWanModel``."""
        if timesteps_B_T.dim() != 2 or timesteps_B_T.shape[1] != 1:
            raise AssertionError("timesteps_B_T must be [B, 1]")  # wan2pt1.py:780
        _reject(slg_layers=slg_layers)
        has_image = self.model_type in ("i2v", "flf2v")
        if has_image and (frame_cond_crossattn_emb_B_L_D is None or y_B_C_T_H_W is None):
            raise AssertionError("i2v / flf2v: frame_cond_crossattn_emb_B_L_D and y_B_C_T_H_W are required")  # wan2pt1.py:783-784
        return self._run(x_B_C_T_H_W, y_B_C_T_H_W, timesteps_B_T[:, 0], crossattn_emb,
                         frame_cond_crossattn_emb_B_L_D if has_image else None, plain_temporal=False)
