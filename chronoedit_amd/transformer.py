"""Drop-in `ChronoEditTransformer3DModel` whose forward runs on the gfx950 HIP kernels.

Mirrors /root/reference/chronoedit_diffusers/transformer_chronoedit.py:298-476:
  * same constructor keyword arguments and `.config` fields (:341-360);
  * same parameter tree with the diffusers state-dict names (key list:
    chronoedit_diffsynth/wan_video_dit_chronoedit.py:439-496), built from real nn.Linear / nn.Conv3d
    / nn.LayerNorm modules so `load_state_dict`, `.to`, PEFT LoRA injection/fusion keep working;
  * same `forward(hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image,
    return_dict, attention_kwargs)` signature, return types and error behaviour (:397-476);
  * same dtype islands: `_keep_in_fp32_modules` parameters stay fp32 (:338).
The modules only HOLD parameters; the arithmetic is `DiTEngine` below, which sequences
libchronoedit_hip.so launches on the current stream (no torch math on the hot path).
"""
from __future__ import annotations

import inspect
import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import ops, weights
from .weights import LoraMixin


_VT_GEMM_SWAPPED = __import__("os").environ.get("CE_VT_GEMM") == "row"  # (tools: the rounds-2-5 form of the V^T product, see forward)


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, d=None):
        return getattr(self, k, d)


class RMSNormParams(nn.Module):
    """Parameter holder for diffusers RMSNorm (norm_q / norm_k / norm_added_k)."""

    def __init__(self, dim: int, eps: float, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))


class AttentionParams(nn.Module):
    """Parameter holder with diffusers `Attention` attribute names (transformer_chronoedit.py:231-258)."""

    def __init__(self, dim: int, heads: int, eps: float, added_kv_proj_dim: Optional[int], device=None, dtype=None):
        super().__init__()
        kw = dict(device=device, dtype=dtype)
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, **kw)
        self.to_k = nn.Linear(dim, dim, **kw)
        self.to_v = nn.Linear(dim, dim, **kw)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim, **kw), nn.Dropout(0.0)])
        self.norm_q = RMSNormParams(dim, eps, **kw)
        self.norm_k = RMSNormParams(dim, eps, **kw)
        self.add_k_proj = self.add_v_proj = self.norm_added_k = None
        if added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, dim, **kw)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, dim, **kw)
            self.norm_added_k = RMSNormParams(dim, eps, **kw)


class _GELUProj(nn.Module):
    def __init__(self, din, dout, device=None, dtype=None):
        super().__init__()
        self.proj = nn.Linear(din, dout, device=device, dtype=dtype)


class FeedForwardParams(nn.Module):
    """diffusers FeedForward layout: net.0.proj, net.1 (Dropout), net.2."""

    def __init__(self, dim, inner, dim_out=None, device=None, dtype=None):
        super().__init__()
        self.net = nn.ModuleList([
            _GELUProj(dim, inner, device=device, dtype=dtype),
            nn.Dropout(0.0),
            nn.Linear(inner, dim_out if dim_out is not None else dim, device=device, dtype=dtype),
        ])


class BlockParams(nn.Module):
    """ChronoEditTransformerBlock parameters (transformer_chronoedit.py:216-265)."""

    def __init__(self, dim, ffn_dim, heads, cross_attn_norm, eps, added_kv_proj_dim, device=None, dtype=None):
        super().__init__()
        self.attn1 = AttentionParams(dim, heads, eps, None, device=device, dtype=dtype)
        self.attn2 = AttentionParams(dim, heads, eps, added_kv_proj_dim, device=device, dtype=dtype)
        # norm1 / norm3 have no affine parameters; norm2 is FP32LayerNorm(affine) kept in fp32 (:338)
        self.norm2 = nn.LayerNorm(dim, eps, elementwise_affine=True, device=device, dtype=torch.float32) if cross_attn_norm else nn.Identity()
        self.ffn = FeedForwardParams(dim, ffn_dim, device=device, dtype=dtype)
        self.scale_shift_table = nn.Parameter(torch.randn(1, 6, dim, device=device, dtype=torch.float32) / dim**0.5)


class _TimestepEmbedding(nn.Module):
    def __init__(self, din, dim, device=None):
        super().__init__()
        self.linear_1 = nn.Linear(din, dim, device=device, dtype=torch.float32)
        self.linear_2 = nn.Linear(dim, dim, device=device, dtype=torch.float32)


class _TextProjection(nn.Module):
    def __init__(self, din, dim, device=None, dtype=None):
        super().__init__()
        self.linear_1 = nn.Linear(din, dim, device=device, dtype=dtype)
        self.linear_2 = nn.Linear(dim, dim, device=device, dtype=dtype)


class _ImageEmbedding(nn.Module):
    def __init__(self, din, dout, device=None, dtype=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(din, device=device, dtype=torch.float32)
        self.ff = FeedForwardParams(din, din, dout, device=device, dtype=dtype)
        self.norm2 = nn.LayerNorm(dout, device=device, dtype=torch.float32)


class ConditionEmbedderParams(nn.Module):
    """ChronoEditTimeTextImageEmbedding parameters (transformer_chronoedit.py:126-145)."""

    def __init__(self, dim, time_freq_dim, time_proj_dim, text_embed_dim, image_embed_dim, device=None, dtype=None):
        super().__init__()
        self.time_embedder = _TimestepEmbedding(time_freq_dim, dim, device=device)
        self.time_proj = nn.Linear(dim, time_proj_dim, device=device, dtype=dtype)
        self.text_embedder = _TextProjection(text_embed_dim, dim, device=device, dtype=dtype)
        self.image_embedder = None
        if image_embed_dim is not None:
            self.image_embedder = _ImageEmbedding(image_embed_dim, dim, device=device, dtype=dtype)


import itertools

_GENERATION = itertools.count(1)


class ChronoEditTransformer3DModel(LoraMixin, nn.Module):
    """MI355X drop-in for the reference class of the same name (transformer_chronoedit.py:298)."""

    _supports_gradient_checkpointing = False
    _no_split_modules = ["BlockParams"]
    _keep_in_fp32_modules = ["time_embedder", "scale_shift_table", "norm1", "norm2", "norm3"]
    _keys_to_ignore_on_load_unexpected = ["norm_added_q"]

    def __init__(
        self,
        patch_size: Tuple[int, int, int] = (1, 2, 2),
        num_attention_heads: int = 40,
        attention_head_dim: int = 128,
        in_channels: int = 16,
        out_channels: int = 16,
        text_dim: int = 4096,
        freq_dim: int = 256,
        ffn_dim: int = 13824,
        num_layers: int = 40,
        cross_attn_norm: bool = True,
        qk_norm: Optional[str] = "rms_norm_across_heads",
        eps: float = 1e-6,
        image_dim: Optional[int] = None,
        added_kv_proj_dim: Optional[int] = None,
        rope_max_seq_len: int = 1024,
        rope_temporal_skip_len: int = 8,
        device=None,
        dtype: torch.dtype = torch.bfloat16,
    ) -> None:
        super().__init__()
        if qk_norm != "rms_norm_across_heads":
            raise NotImplementedError("only qk_norm='rms_norm_across_heads' (the ChronoEdit/Wan setting) is supported")
        if tuple(patch_size) != (1, 2, 2):
            raise NotImplementedError("patch_size must be (1, 2, 2)")
        if attention_head_dim != 128:
            raise NotImplementedError("the gfx950 attention kernel is specialised for head_dim 128")
        self.config = _Config(
            patch_size=tuple(patch_size), num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
            in_channels=in_channels, out_channels=out_channels or in_channels, text_dim=text_dim, freq_dim=freq_dim,
            ffn_dim=ffn_dim, num_layers=num_layers, cross_attn_norm=cross_attn_norm, qk_norm=qk_norm, eps=eps,
            image_dim=image_dim, added_kv_proj_dim=added_kv_proj_dim, rope_max_seq_len=rope_max_seq_len,
            rope_temporal_skip_len=rope_temporal_skip_len,
        )
        inner = num_attention_heads * attention_head_dim
        kw = dict(device=device, dtype=dtype)
        self.patch_embedding = nn.Conv3d(in_channels, inner, kernel_size=patch_size, stride=patch_size, **kw)
        self.condition_embedder = ConditionEmbedderParams(inner, freq_dim, inner * 6, text_dim, image_dim, **kw)
        self.blocks = nn.ModuleList(
            [BlockParams(inner, ffn_dim, num_attention_heads, cross_attn_norm, eps, added_kv_proj_dim, **kw) for _ in range(num_layers)]
        )
        self.proj_out = nn.Linear(inner, self.config.out_channels * math.prod(patch_size), **kw)
        self.scale_shift_table = nn.Parameter(torch.randn(1, 2, inner, device=device, dtype=torch.float32) / inner**0.5)
        self._engine: Optional["DiTEngine"] = None
        self._gen = next(_GENERATION)  # engine_generation(): see there
        self.cache_context = False  # reuse K3/K13 results while the conditioning tensors are unchanged
        self.gemm_dtype = "bf16"    # "fp8": the six large Linears of every block on the OCP-e4m3 MX matrix path
        self.fp8_linears = self.FP8_LINEARS  # which of them, once the fp8 mode is on (enable_fp8_gemms(linears= / policy=))
        self.attn_dtype = "bf16"    # "mxfp8": self-attention on the MX-fp8 matrix instruction (csrc/ce_attn_fp8.hip)
        self.v_transposed = True    # bf16 self-attention takes V^T straight from the projection (swapped GEMM) and stages it by LDS-DMA
        self.sp_batch_cfg = True    # sequence-parallel forwards take the guidance pair as one batch of two (blocked-layout kernels)
        self.fp8_fuse_quant = True  # MX fp8 mode: the FFN-up epilogue emits the FFN-down operand quantised (no bf16 hidden matrix, no quant pass)
        self.fp8_fuse_attn_quant = True  # ... and both attention kernels emit the out-projections' operands (A/B switch; needs fp8_fuse_quant)
        self.cross_vt = True        # cross-attention takes V^T of the text / image context straight from the context projections (LDS-DMA kernel)
        self._sp = None             # Ulysses sequence parallelism (chronoedit_amd.parallel), off by default
        self._cfgp = None           # CFG parallelism on top of it (two Ulysses groups), off by default

    # -- reference-compatible helpers --------------------------------------------------
    @property
    def dtype(self) -> torch.dtype:
        return self.proj_out.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.proj_out.weight.device

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype: torch.dtype = torch.bfloat16,
                        device=None, **unused) -> "ChronoEditTransformer3DModel":
        """diffusers-layout checkpoint directory -> model (run_inference_diffusers.py:349-353): ``config.json`` gives the
        constructor arguments, the safetensors shard(s) the parameters; fp32 islands stay fp32 (``_keep_in_fp32_modules``)
        and ``norm_added_q`` keys are ignored (transformer_chronoedit.py:338-339)."""
        cfg = weights.read_config(path, subfolder)
        known = set(inspect.signature(cls.__init__).parameters) - {"self", "device", "dtype"}
        model = cls(**{k: v for k, v in cfg.items() if k in known}, device=device, dtype=torch_dtype)
        sd = weights.load_state_dict_files(weights.shard_files(path, subfolder))
        weights.assign_state_dict(model, sd, ignore_unexpected=cls._keys_to_ignore_on_load_unexpected)
        model.invalidate()
        return model

    def load_wan_native_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Load a checkpoint in the original Wan naming used by the reference's two sibling stacks (diffsynth ``WanModel``,
        ``_src`` ``wan2pt1``; correspondence: wan_video_dit_chronoedit.py:434-505)."""
        renamed = weights.wan_native_to_diffusers(sd, [k for k, _ in self.named_parameters()])
        weights.assign_state_dict(self, renamed, ignore_unexpected=self._keys_to_ignore_on_load_unexpected)
        self.invalidate()
        return self

    def wan_native_state_dict(self) -> Dict[str, torch.Tensor]:
        """The parameters under their Wan-native names (e.g. to hand LoRA-fused weights to the sibling stacks)."""
        return {weights.diffusers_to_wan_native_key(k): p.detach() for k, p in self.named_parameters()}

    def save_pretrained(self, path: str, max_shard_bytes: int = 5 << 30):
        """config.json + safetensors shard(s) in the layout from_pretrained reads."""
        return weights.save_pretrained(self, path, dict(vars(self.config)), max_shard_bytes, "ChronoEditTransformer3DModel")

    def load_synthetic_(self, params: Dict[str, torch.Tensor]):
        """Copy a {diffusers key: tensor} dict (e.g. oracle.make_synthetic_params) into the tree."""
        own = dict(self.named_parameters())
        missing = set(own) - set(params)
        extra = set(params) - set(own)
        if missing or extra:
            raise KeyError(f"state mismatch: missing {sorted(missing)[:4]} unexpected {sorted(extra)[:4]}")
        with torch.no_grad():
            for k, p in own.items():
                p.copy_(params[k].to(p.dtype))
        self.invalidate()
        return self

    def enable_sequence_parallel(self, group=None, force: bool = False, owned_comm: bool = False):
        """Shard the token axis over the ranks of `group` (Ulysses: three all-to-all per self-attention).  Every rank
        must call forward with the same (replicated) inputs and receives the full output.  force, owned_comm: see parallel.Ulysses
        (owned_comm = the exchanges on a RCCL communicator owned by libchronoedit_hip: the sharded step becomes hipGraph-capturable)."""
        from .parallel import Ulysses
        self._sp = Ulysses(group, force=force, owned_comm=owned_comm)
        if self.config.num_attention_heads % self._sp.world:
            raise ValueError(f"{self.config.num_attention_heads} heads do not divide over {self._sp.world} ranks")
        if self._engine is not None:
            self._engine._ws = {}  # workspaces carry the exchange buffers of the group
            self._engine.ws_generation += 1
        return self

    def enable_cfg_parallel(self):
        """World of W = 2 S ranks: the conditional / unconditional passes of a guidance step run side by side on two S-way
        Ulysses groups (parallel.CFGParallel); `pipeline.denoise_step` exchanges the two predictions.  Collective: every
        rank of the world must call it."""
        from .parallel import CFGParallel, Ulysses
        self._cfgp = CFGParallel()
        self._sp = Ulysses(self._cfgp.sp_group)
        if self.config.num_attention_heads % self._sp.world:
            raise ValueError(f"{self.config.num_attention_heads} heads do not divide over {self._sp.world} ranks")
        if self._engine is not None:
            self._engine._ws = {}
            self._engine.ws_generation += 1
        return self

    def clear_context_cache(self):
        if self._engine is not None:
            self._engine.clear_context_cache()

    @torch.no_grad()
    def prime_context(self, encoder_hidden_states: torch.Tensor, encoder_hidden_states_image: Optional[torch.Tensor]):
        """Compute the step-invariant conditioning work for exactly these tensors now, so that the next forward
        that receives them - e.g. the one a hipGraph capture records - finds the cache entry (only with `cache_context`)."""
        if self.cache_context:
            self.engine()._context(encoder_hidden_states, encoder_hidden_states_image)

    def invalidate(self):
        """Call after changing parameters in place (LoRA fuse, load_state_dict): re-packs on next forward."""
        self._engine = None

    FP8_LINEARS = ("qkv", "o1", "q2", "o2", "f1", "f2")  # fused q|k|v, self-attention out, cross-attention q, cross-attention out, FFN up, FFN down
    # Mixed-precision policies of the fp8 mode (round 6): which of the six large Linears of a block run on the MX fp8 GEMM - the others stay
    # on the bf16 GEMM.  From the measurement in profiles/r06_fp8_sensitivity.txt (one full-width block at N = 7 200 against the fp32 oracle,
    # one Linear in fp8 at a time): the UNGATED cross-attention out-projection `o2` alone carries 3.3e-2 of the 3.7e-2 the fp8 mode adds to
    # the bf16 path's 4.8e-3 (its result enters the residual stream at full scale; `o1` and `f2` are multiplied by their AdaLN gates first,
    # `qkv` feeds a softmax); the MXFP8 self-attention adds 1.0e-3.  "fast" = all six (7.8 x the bf16 error per block); "accurate" = all but
    # `o2` (4.0 x) for one 5120 x 5120 GEMM per block back on the bf16 kernel.
    FP8_POLICIES = {"fast": ("qkv", "o1", "q2", "o2", "f1", "f2"), "accurate": ("qkv", "o1", "q2", "f1", "f2")}

    def enable_fp8_gemms(self, on: bool = True, mx: bool = True, linears=None, policy: Optional[str] = None):
        """BASELINE.json configs[4]: run the six large Linears of every block (fused q|k|v, the two output projections, the
        cross-attention query, FFN up / down) in fp8 e4m3 with fp32 accumulation on the MX matrix instruction.
        mx=True (default since round 4): OCP MXFP8 operands - one E8M0 scale per 32 consecutive input channels of every activation row and
        of every weight row, applied inside the matrix pipe (`ce_gemm_mxfp8`; quantisers `ce_quant_rows_mxfp8` / `ce_ln_affine_mxfp8`);
        mx=False: one fp32 scale per token row / per output channel (`ce_gemm_fp8`, the round-1..3 contract).  Weights are quantised once;
        attention, norms, the residual stream, the conditioning projections and the head stay bf16 / fp32.  The bf16 parameters are kept.
        linears / policy (round 6): a subset of FP8_LINEARS, or the name of one of FP8_POLICIES ("fast" = all six, the default; "accurate" = all but the
        ungated cross-attention out-projection, 4.0 x instead of 7.8 x the bf16 path's error per block) - Linears not named run the bf16 GEMM on bf16 activations."""
        if policy is not None:
            if linears is not None:
                raise ValueError("enable_fp8_gemms: give `linears` or `policy`, not both")
            if policy not in self.FP8_POLICIES:
                raise ValueError(f"enable_fp8_gemms: unknown policy {policy!r} (one of {sorted(self.FP8_POLICIES)})")
            linears = self.FP8_POLICIES[policy]
        linears = tuple(self.FP8_LINEARS if linears is None else linears)
        bad = [n for n in linears if n not in self.FP8_LINEARS]
        if bad:
            raise ValueError(f"enable_fp8_gemms: unknown Linear name(s) {bad} (of {self.FP8_LINEARS})")
        self.fp8_linears = tuple(n for n in self.FP8_LINEARS if n in linears)
        self.gemm_dtype = ("mxfp8" if mx else "fp8") if (on and self.fp8_linears) else "bf16"
        self._engine = None
        return self

    def enable_transposed_v(self, on: bool = True):
        """bf16 self-attention operand form (default on): the V third of the fused projection is taken as V^T = W_v.h^T (the GEMM
        with its operand roles swapped and the bias along rows, CE_EPI_BIAS_ROW) so that K and V^T tiles both reach the attention
        kernel's LDS by LDS-DMA (`ce_attention_vt_bf16`: +4.4 % on the kernel, no transpose pass).  Off = the fused q|k|v GEMM and
        the register-staged kernel; same arithmetic up to the summation order inside a key tile."""
        self.v_transposed = bool(on)
        if self._engine is not None:
            self._engine.v_transposed = self.v_transposed
            self._engine._ws = {}
            self._engine.ws_generation += 1
        return self

    def enable_cross_vt(self, on: bool = True):
        """Cross-attention operand form (default on): the V halves of the context projections are produced as V^T (the swapped-role GEMM,
        all layers in one launch per context stream) and both segments' K / V^T tiles reach the kernel's LDS by LDS-DMA
        (`ce_attention_2seg_vt_bf16`).  Off = row-major V and the register-staged two-segment kernel; same arithmetic up to the
        summation order inside a key tile."""
        self.cross_vt = bool(on)
        self._engine = None
        return self

    def enable_fp8_attention(self, on: bool = True, cross: bool = False):
        """BASELINE.json configs[4] "fp8 weights+attn": the attention of every block under the MXFP8 contract of
        csrc/ce_attn_fp8.hip - q / k (after RMSNorm + RoPE) in e4m3 with one E8M0 scale per 32 head channels, v per 32 keys,
        Q.K^T and P.V on v_mfma_scale_f32_32x32x64_f8f6f4, P in e4m3, fp32 accumulation.  cross (round 5): the cross-attention too - the
        text and the image segment each as one MXFP8 attention over the context's quantised K / V^T (made once per context: cached per
        edit), the image segment adding the text segment's bf16 result (`ce_attention_mxfp8_add`).  OFF by default - measured on MI355X at
        the full width (tools/fp8_cross_probe.py, profiles/r05_fp8_cross_attention_probe.txt): with 512 + 257 keys the two MXFP8 launches
        are prologue / epilogue bound (0.199 + 0.209 ms + 0.06 ms more producer work against 0.315 ms for the bf16 two-segment kernel with
        the fused MX output) and the block's error against fp32 rises from 7.85 x to 8.75 x the bf16 path's: slower AND less accurate, so
        the default keeps the 3 % of the attention flops that are cross-attention in bf16.  Everything else is unchanged; independent of
        enable_fp8_gemms."""
        self.attn_dtype = "mxfp8" if on else "bf16"
        self.fp8_cross = bool(on and cross)
        self._engine = None  # (the context operands and the workspaces depend on it)
        return self

    def attention_path(self) -> str:
        """The self-attention arithmetic the next forward will actually run: "mxfp8" only when enable_fp8_attention() is on AND the
        tokens are not sharded - the sequence-parallel path exchanges bf16 q / k / v and runs the bf16 kernel (bench.py labels
        its lines from this, not from the flag)."""
        sharded = self._sp is not None and self._sp.sharded
        return "mxfp8" if (self.attn_dtype == "mxfp8" and not sharded) else "bf16"

    def _apply(self, fn, *a, **kw):  # .to() / .cuda() / .cpu() re-create storages
        self._engine = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._engine = None
        return super().load_state_dict(*a, **kw)

    def engine(self) -> "DiTEngine":
        if self._engine is None:
            self._engine = DiTEngine(self)
            self._gen = next(_GENERATION)
        return self._engine

    def engine_generation(self):
        """A value that changes whenever something a captured step depends on may have been re-created lazily: the engine itself (packed
        weights), its workspaces, or a mode that alters the launch sequence (sequence / CFG parallelism, operand forms, fp8 switches, the
        RoPE spelling).  `pipeline.denoise` keys its "this shape has already run eagerly once" set on it, so a capture never records a
        first-time initialisation.  (Process-unique, unlike id(): a recycled address cannot alias an earlier model.)"""
        eng = self._engine
        sp = self._sp
        return (self._gen, None if eng is None else eng.ws_generation, self.gemm_dtype, self.fp8_linears, self.attn_dtype, self.v_transposed, self.cross_vt,
                self.sp_batch_cfg, bool(getattr(self, "rope_plain_temporal", False)), self.cache_context,
                None if sp is None else (sp.world, sp.rank), self._cfgp is not None)

    @torch.no_grad()
    def forward(
        self,
        hidden_states: torch.Tensor,
        timestep: torch.LongTensor,
        encoder_hidden_states: torch.Tensor,
        encoder_hidden_states_image: Optional[torch.Tensor] = None,
        return_dict: bool = True,
        attention_kwargs: Optional[Dict[str, Any]] = None,
    ) -> Union[Transformer2DModelOutput, Tuple[torch.Tensor]]:
        if not hidden_states.is_cuda:
            raise ops.HipKernelError("ChronoEditTransformer3DModel (chronoedit_amd) runs only on an MI355X device: "
                                     "there is no CPU fallback (use oracle/ for CPU reference numbers)")
        B = hidden_states.shape[0]
        ts = timestep.reshape(-1)
        if ts.numel() == 1 and B > 1:
            ts = ts.expand(B)
        output = self.engine().forward(hidden_states, ts, encoder_hidden_states, encoder_hidden_states_image).to(hidden_states.dtype)
        if not return_dict:
            return (output,)
        return Transformer2DModelOutput(sample=output)


# ------------------------------------------------------------------------------------------
def rope_cos_sin(head_dim: int, max_len: int, skip_len: int, T: int, Hp: int, Wp: int, theta: float = 10000.0,
                 plain_temporal: bool = False) -> torch.Tensor:
    """[T*Hp*Wp, head_dim/2, 2] fp32 (cos, sin) of ChronoEditRotaryPosEmbed
    (transformer_chronoedit.py:168-213): per-axis dims (t, h, w) = (hd - 4*(hd//6), 2*(hd//6), 2*(hd//6)),
    angles in fp64; temporal indices {0, skip_len-1} when T == 2 (:205-207).  plain_temporal: indices 0..T-1 for any T,
    what the diffsynth call path uses (wan_video_new_chronoedit.py:1428-1432)."""
    assert plain_temporal or T == 2 or T == skip_len, f"num_frames must be 2 or {skip_len}, but got {T}"
    h_dim = w_dim = 2 * (head_dim // 6)
    t_dim = head_dim - h_dim - w_dim

    def ang(dim, idx):
        f = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        return torch.outer(idx.to(torch.float64), f)

    t_idx = torch.tensor([0, skip_len - 1]) if (T == 2 and not plain_temporal) else torch.arange(T)
    assert max(T, Hp, Wp, skip_len) <= max_len
    a_t = ang(t_dim, t_idx).view(T, 1, 1, -1).expand(T, Hp, Wp, -1)
    a_h = ang(h_dim, torch.arange(Hp)).view(1, Hp, 1, -1).expand(T, Hp, Wp, -1)
    a_w = ang(w_dim, torch.arange(Wp)).view(1, 1, Wp, -1).expand(T, Hp, Wp, -1)
    a = torch.cat([a_t, a_h, a_w], dim=-1).reshape(T * Hp * Wp, head_dim // 2)
    return torch.stack([torch.cos(a), torch.sin(a)], dim=-1).to(torch.float32).contiguous()


def _pad_k(w: torch.Tensor, mult: int = 64) -> torch.Tensor:
    K = w.shape[1]
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return w.contiguous()
    out = torch.zeros((w.shape[0], Kp), dtype=w.dtype, device=w.device)
    out[:, :K] = w
    return out


def _pad_n(w: torch.Tensor, b: Optional[torch.Tensor], mult: int = 8):
    N = w.shape[0]
    Np = (N + mult - 1) // mult * mult
    if Np == N:
        return w, b
    w2 = torch.zeros((Np, w.shape[1]), dtype=w.dtype, device=w.device)
    w2[:N] = w
    b2 = None
    if b is not None:
        b2 = torch.zeros((Np,), dtype=b.dtype, device=b.device)
        b2[:N] = b
    return w2, b2


class DiTEngine:
    """Packs the parameters once and runs the forward as a fixed sequence of HIP launches."""

    def __init__(self, model: ChronoEditTransformer3DModel):
        cfg = model.config
        self.cfg = cfg
        self.model = model
        dev = model.device
        if dev.type != "cuda":
            raise ops.HipKernelError("DiTEngine needs the model on the GPU")
        if model.dtype != torch.bfloat16:
            raise TypeError("the HIP path computes in bf16 (run_inference_diffusers.py hard-codes bf16, :344-362); "
                            f"got model dtype {model.dtype}")
        self.dev = dev
        self.D = cfg.num_attention_heads * cfg.attention_head_dim
        self.H = cfg.num_attention_heads
        self.F = cfg.ffn_dim
        self.L = cfg.num_layers
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        ce = model.condition_embedder

        # K1 patch embedding as a GEMM over im2col columns; K = C*4 zero-padded to 64
        self.kpatch = (cfg.in_channels * 4 + 63) // 64 * 64
        self.w_patch = _pad_k(model.patch_embedding.weight.detach().reshape(self.D, -1))
        self.b_patch = f32(model.patch_embedding.bias)

        # K2
        self.te_w1, self.te_b1 = ce.time_embedder.linear_1.weight.detach().contiguous(), f32(ce.time_embedder.linear_1.bias)
        self.te_w2, self.te_b2 = ce.time_embedder.linear_2.weight.detach().contiguous(), f32(ce.time_embedder.linear_2.bias)
        self.tp_w, self.tp_b = ce.time_proj.weight.detach().contiguous(), f32(ce.time_proj.bias)

        # K3
        self.tx_w1, self.tx_b1 = _pad_k(ce.text_embedder.linear_1.weight.detach()), f32(ce.text_embedder.linear_1.bias)
        self.tx_w2, self.tx_b2 = ce.text_embedder.linear_2.weight.detach().contiguous(), f32(ce.text_embedder.linear_2.bias)
        self.has_image = ce.image_embedder is not None
        if self.has_image:
            ie = ce.image_embedder
            self.im_n1 = (f32(ie.norm1.weight), f32(ie.norm1.bias), ie.norm1.eps)
            self.im_w1, self.im_b1 = _pad_k(ie.ff.net[0].proj.weight.detach()), f32(ie.ff.net[0].proj.bias)
            self.im_w2, self.im_b2 = _pad_k(ie.ff.net[2].weight.detach()), f32(ie.ff.net[2].bias)
            self.im_n2 = (f32(ie.norm2.weight), f32(ie.norm2.bias), ie.norm2.eps)

        # per-block packs.  Fused QKV / KV weights are single buffers; the module parameters are
        # re-pointed at views of them so the 14B model keeps ONE copy of every weight in HBM.
        self.blk = []
        tables = []
        for blk in model.blocks:
            p = SimpleNamespace()
            a1, a2 = blk.attn1, blk.attn2
            p.w_qkv = self._fuse([a1.to_q, a1.to_k, a1.to_v])
            p.b_qkv = torch.cat([f32(a1.to_q.bias), f32(a1.to_k.bias), f32(a1.to_v.bias)])
            p.nq1, p.nk1 = f32(a1.norm_q.weight), f32(a1.norm_k.weight)
            p.w_o1, p.b_o1 = a1.to_out[0].weight.detach().contiguous(), f32(a1.to_out[0].bias)
            p.w_q2, p.b_q2 = a2.to_q.weight.detach().contiguous(), f32(a2.to_q.bias)
            p.nq2, p.nk2 = f32(a2.norm_q.weight), f32(a2.norm_k.weight)
            p.has_img = a2.add_k_proj is not None
            if p.has_img:
                p.nk_i = f32(a2.norm_added_k.weight)
            p.w_o2, p.b_o2 = a2.to_out[0].weight.detach().contiguous(), f32(a2.to_out[0].bias)
            if cfg.cross_attn_norm:
                p.n2w, p.n2b = f32(blk.norm2.weight), f32(blk.norm2.bias)
            else:
                p.n2w = p.n2b = None
            p.w_f1, p.b_f1 = blk.ffn.net[0].proj.weight.detach().contiguous(), f32(blk.ffn.net[0].proj.bias)
            p.w_f2, p.b_f2 = blk.ffn.net[2].weight.detach().contiguous(), f32(blk.ffn.net[2].bias)
            tables.append(f32(blk.scale_shift_table).reshape(6, self.D))
            self.blk.append(p)
        self.fp8_attn = model.attn_dtype == "mxfp8"
        self.fp8_cross = self.fp8_attn and bool(getattr(model, "fp8_cross", False))  # cross-attention under the MXFP8 contract too
        self.fp8 = model.gemm_dtype in ("fp8", "mxfp8")
        self.mx = model.gemm_dtype == "mxfp8"  # MX block scales on both GEMM operands (ce_gemm_mxfp8)
        self.fp8_set = frozenset(getattr(model, "fp8_linears", model.FP8_LINEARS)) if self.fp8 else frozenset()  # which Linears (mixed precision)
        self.fuse_quant = bool(getattr(model, "fp8_fuse_quant", True)) and self.F % 128 == 0  # MX: quantisation fused into the FFN-up epilogue
        self.fuse_attn_quant = self.fuse_quant and bool(getattr(model, "fp8_fuse_attn_quant", True))
        self.v_transposed = bool(getattr(model, "v_transposed", True))
        if self.fp8:
            if self.D % 256 or self.F % 256:
                raise NotImplementedError("fp8 GEMMs need inner and ffn dims that are multiples of 256")
            for p in self.blk:  # per-output-channel e4m3 copies of the six large weights (the bf16 originals stay)
                for name in sorted(self.fp8_set):
                    w = getattr(p, "w_" + name)  # (MX: the weight operand's scale order, ce_quant_rows_mxfp8_w)
                    setattr(p, "q_" + name, ops.quant_rows_mxfp8(w, w_order=True) if self.mx else ops.quant_rows_fp8(w))
        # K13 for ALL layers as one GEMM per context stream and operand: the per-layer to_k (to_v, add_k_proj, add_v_proj) weights are
        # re-homed, layer after layer, in one [L*D, D] buffer each (the step-invariant projections of 769 context rows are 160 small
        # GEMMs otherwise: 0.5-1.0 PFLOP/s at M = 514 / 1024 against 1.35 for one [M, L*D] product).  K and V apart (round 4): the V
        # half is taken as V^T = W_v . ctx^T - the same GEMM with its operand roles swapped and the bias along rows - which is the
        # operand the LDS-DMA form of the cross-attention kernel wants, at no extra pass.
        self.w_k_t_all = self._fuse([blk.attn2.to_k for blk in model.blocks])
        self.b_k_t_all = torch.cat([f32(blk.attn2.to_k.bias) for blk in model.blocks])
        self.w_v_t_all = self._fuse([blk.attn2.to_v for blk in model.blocks])
        self.b_v_t_all = torch.cat([f32(blk.attn2.to_v.bias) for blk in model.blocks])
        self.all_img = all(p.has_img for p in self.blk)  # (added_kv_proj_dim is one constructor argument: all layers or none)
        if self.all_img:
            self.w_k_i_all = self._fuse([blk.attn2.add_k_proj for blk in model.blocks])
            self.b_k_i_all = torch.cat([f32(blk.attn2.add_k_proj.bias) for blk in model.blocks])
            self.w_v_i_all = self._fuse([blk.attn2.add_v_proj for blk in model.blocks])
            self.b_v_i_all = torch.cat([f32(blk.attn2.add_v_proj.bias) for blk in model.blocks])
        self.cross_vt = bool(getattr(model, "cross_vt", True))
        self._ctx_bufs = {}
        self.tables = torch.stack(tables, 0).contiguous()  # [L, 6, D]
        self.table_out = f32(model.scale_shift_table).reshape(1, 2, self.D)
        self.w_out, self.b_out = _pad_n(model.proj_out.weight.detach().contiguous(), f32(model.proj_out.bias))
        self.ones = torch.ones(self.D, dtype=torch.float32, device=dev)
        self.zeros = torch.zeros(self.D, dtype=torch.float32, device=dev)
        self._rope = {}
        self._ws = {}
        self.ws_generation = 0  # bumped when a workspace is evicted (a shape seen before is then "new" again)
        self._ctx_key = None
        self._ctx = None
        self._ctx_refs = None

    def _fuse(self, linears) -> torch.Tensor:
        """cat the [out,in] weights into one buffer and re-point the module parameters at its rows."""
        w = torch.cat([l.weight.detach() for l in linears], dim=0).contiguous()
        o = 0
        for l in linears:
            n = l.weight.shape[0]
            l.weight.data = w[o : o + n]
            o += n
        return w

    def _ln_linear(self, ws, x: torch.Tensor, a_row, b_row, p, name: str, out: torch.Tensor, ab_rows: int = 0, ab_stride: int = 0, **kw):
        """LayerNorm (+ affine / AdaLN rows) followed by one of the large projections.  fp8 mode: the LN kernel emits the fp8
        operand and its row scales directly (no bf16 round trip through HBM, no separate quantisation pass)."""
        eps = self.cfg.eps
        if name not in self.fp8_set:
            ops.ln_affine(x, a_row, b_row, eps, out=ws.h, ab_rows=ab_rows, ab_stride=ab_stride)
            return ops.gemm(ws.h, getattr(p, "w_" + name), getattr(p, "b_" + name), out=out, **kw)
        aq = ws.a8[:, : x.shape[1]]
        wq, sw = getattr(p, "q_" + name)
        if self.mx:
            ops.ln_affine_mxfp8(x, a_row, b_row, eps, out=aq, scale=ws.s8, ab_rows=ab_rows, ab_stride=ab_stride)
            return ops.gemm_mxfp8(aq, ws.s8, wq, sw, getattr(p, "b_" + name), out=out, **kw)
        ops.ln_affine_fp8(x, a_row, b_row, eps, out=aq, scale=ws.s8, ab_rows=ab_rows, ab_stride=ab_stride)
        return ops.gemm_fp8(aq, ws.s8, wq, sw, getattr(p, "b_" + name), out=out, **kw)

    def _linear(self, ws, a: torch.Tensor, p, name: str, out: torch.Tensor, quantised: bool = False, **kw):
        """One of the six large projections of a block: bf16 GEMM, or (fp8 mode) row-quantise the activations and run the MX GEMM.
        quantised: the producer already wrote the MX operand into ws.a8 / ws.s8 (an attention kernel's fused output)."""
        w, b = getattr(p, "w_" + name), getattr(p, "b_" + name)
        if name not in self.fp8_set:
            assert not quantised, name  # (a bf16 Linear takes bf16 activations: the caller did not ask its producer for the MX operand)
            return ops.gemm(a, w, b, out=out, **kw)
        K = a.shape[1]
        aq = ws.a8[:, :K]
        wq, sw = getattr(p, "q_" + name)
        if self.mx:
            if not quantised:
                ops.quant_rows_mxfp8(a, out=aq, scale=ws.s8)
            return ops.gemm_mxfp8(aq, ws.s8, wq, sw, b, out=out, **kw)
        ops.quant_rows_fp8(a, out=aq, scale=ws.s8)
        return ops.gemm_fp8(aq, ws.s8, wq, sw, b, out=out, **kw)

    def _self_attention_ulysses(self, ws, sp, x, a_row, b_row, p, cs, N: int, Nl: int, B: int = 1):
        """Self-attention with the tokens sharded over sp.world ranks (parallel.py has the layout story); B samples stacked [sample][local
        token] along the rows (B > 1: the blocked receive layout, Nl a multiple of 64).
        The fused q|k|v projection runs as two GEMMs - [k | v] first - so that the k|v exchange (2/3 of the volume) is on
        the wire while the q projection is still computing; q / k are normalised and rotated by the pass that writes the
        all-to-all send layout; the attention kernel and the out-projection read the receive buffers in place."""
        D, H, hd, eps, W = self.D, self.H, self.cfg.attention_head_dim, self.cfg.eps, sp.world
        Dl = D // W
        if "qkv" not in self.fp8_set:
            ops.ln_affine(x, a_row, b_row, eps, out=ws.h, ab_rows=Nl, ab_stride=6 * D)
            lin = lambda lo, hi, out: ops.gemm(ws.h, p.w_qkv[lo:hi], p.b_qkv[lo:hi], out=out)
        else:
            aq = ws.a8[:, :D]
            wq, sw = p.q_qkv
            if self.mx:  # (row slices of the fused weight at multiples of D = whole 128-row scale blocks: 512 (D / 128) bytes per block)
                ops.ln_affine_mxfp8(x, a_row, b_row, eps, out=aq, scale=ws.s8, ab_rows=Nl, ab_stride=6 * D)
                sblk = (D // 128) * 512
                lin = lambda lo, hi, out: ops.gemm_mxfp8(aq, ws.s8, wq[lo:hi], sw[lo // 128 * sblk: hi // 128 * sblk], p.b_qkv[lo:hi], out=out)
            else:
                ops.ln_affine_fp8(x, a_row, b_row, eps, out=aq, scale=ws.s8, ab_rows=Nl, ab_stride=6 * D)
                lin = lambda lo, hi, out: ops.gemm_fp8(aq, ws.s8, wq[lo:hi], sw[lo:hi], p.b_qkv[lo:hi], out=out)
        lin(D, 3 * D, ws.qkv[:, D:])
        ops.rope_scatter(ws.qkv, (D, 2 * D), (p.nk1, None), D, W, cs, hd, eps, out=ws.send_kv)
        _, wait_kv = sp.all_to_all(ws.send_kv, ws.recv_kv, async_op=True)
        lin(0, D, ws.qkv[:, :D])
        ops.rope_scatter(ws.qkv, (0,), (p.nq1,), D, W, cs, hd, eps, out=ws.send_q)
        _, wait_q = sp.all_to_all(ws.send_q, ws.recv_q, async_op=True)
        wait_kv.wait()
        wait_q.wait()
        kv = sp.gathered_view(ws.recv_kv)  # [W*Nl = global token, k | v of this rank's heads]
        if B > 1:  # rows are [source rank][sample][local token]: the blocked forms of the transposer and of the attention kernel
            vt_shape = (Dl, B * ops.vt_columns(N))
            ws.vt_sp = ops.v_transpose_blocked(kv[:, Dl:], H // W, B, Nl, N, out=ws.vt_sp if getattr(ws, "vt_sp", torch.empty(0)).shape == vt_shape else None)
            ops.attention_vt_blocked(sp.gathered_view(ws.recv_q), kv[:, :Dl], ws.vt_sp, H // W, B, Nl, N, out=ws.att_g)
        elif self.v_transposed:  # the LDS-DMA form of the kernel wants V^T: one small transpose pass over this rank's heads
            if getattr(ws, "vt_sp", None) is None or ws.vt_sp.shape != (Dl, ops.vt_columns(N)):
                ws.vt_sp = torch.zeros((Dl, ops.vt_columns(N)), dtype=torch.bfloat16, device=self.dev)
            ops.v_transpose(kv[:N, Dl:], H // W, out=ws.vt_sp)
            ops.attention_vt(sp.gathered_view(ws.recv_q), kv[:N, :Dl], ws.vt_sp, H // W, out=ws.att_g)
        else:
            ops.attention(sp.gathered_view(ws.recv_q), kv[:N, :Dl], kv[:N, Dl:], H // W, out=ws.att_g)
        y, _ = sp.all_to_all(ws.att_g.view(W, B * Nl, Dl), ws.att_seg)  # [head group][local row][Dl]
        if "o1" in self.fp8_set:  # the row quantiser wants plain rows: secondary mode, one gather pass
            ws.att.copy_(sp.merge_heads_reference(y))
            return ws.att
        return y  # K-segmented A operand of the out-projection (ce_gemm_aseg_bf16)

    # -- workspaces --------------------------------------------------------------------
    def _workspace(self, N: int):
        ws = self._ws.get(N)
        if ws is None:
            D, F, dev = self.D, self.F, self.dev
            e = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=dev)
            ws = SimpleNamespace(x=e(N, D), h=e(N, D), qkv=e(N, 3 * D), att=e(N, D), q2=e(N, D), ffn=e(N, F),
                                 cols=e(N, self.kpatch), head=e(N, self.w_out.shape[0]))
            if self.fp8:  # activation rows as fp8 + one scale per row (MX: one E8M0 byte per 32 elements, tiled: ops.mx_scale_bytes)
                ws.a8 = torch.empty((N, max(D, F)), dtype=torch.uint8, device=dev)
                ws.s8 = (torch.empty((ops.mx_scale_bytes(N, max(D, F)),), dtype=torch.uint8, device=dev) if self.mx else
                         torch.empty((N,), dtype=torch.float32, device=dev))
                if self.mx and self.fuse_quant and F % 128 == 0:  # the FFN hidden activation as the up-projection's epilogue writes it
                    ws.a8b = torch.empty((N, F), dtype=torch.uint8, device=dev)
                    ws.s8b = torch.empty((ops.mx_scale_bytes(N, F),), dtype=torch.uint8, device=dev)
            if self.fp8_attn:  # MXFP8 q / k (+ E8M0 scale bytes); the V^T tiles depend on the batch split and are sized in forward
                u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=dev)
                ws.q8, ws.k8, ws.sq, ws.sk = u8(N, D), u8(N, D), u8(N, D // 32), u8(N, D // 32)
                ws.v8t = ws.sv = None
            sp = self.model._sp
            if sp is not None and sp.sharded:  # Ulysses exchange buffers (chronoedit_amd/parallel.py): N = local rows
                W, Dl = sp.world, D // sp.world
                ws.send_kv, ws.recv_kv = e(W, N, 2, Dl), e(W, N, 2, Dl)
                ws.send_q, ws.recv_q = e(W, N, 1, Dl), e(W, N, 1, Dl)
                ws.att_g, ws.att_seg = e(W * N, Dl), e(W, N, Dl)
            # keep the two most recent shapes resident: a temporal-reasoning edit alternates between its 8-frame and 2-frame shapes, and a
            # hipGraph capture of a shape seen before must find its workspace (nothing may be allocated for the engine under capture)
            if len(self._ws) > 1:
                self.ws_generation += 1  # the oldest shape leaves: a warm-set entry for it is stale
            keep = list(self._ws.items())[-1:]
            self._ws = dict(keep + [(N, ws)])
        return ws

    def _rope_table(self, T, Hp, Wp):
        plain = bool(getattr(self.model, "rope_plain_temporal", False))
        key = (T, Hp, Wp, plain)
        if key not in self._rope:
            c = self.cfg
            if len(self._rope) >= 4:  # a handful of latent shapes per process (8 / 2 frames, + their sharded slices)
                self._rope.pop(next(iter(self._rope)))
            self._rope[key] = rope_cos_sin(c.attention_head_dim, c.rope_max_seq_len, c.rope_temporal_skip_len, T, Hp, Wp,
                                           plain_temporal=plain).to(self.dev)
        return self._rope[key]

    def is_warm(self, B: int, T: int, Hh: int, Ww: int) -> bool:
        """Has a forward of exactly this shape already run through this engine (workspace and RoPE table resident)?  What a hipGraph
        capture without a warm-up step requires (pipeline.GraphedDenoiser)."""
        plain = bool(getattr(self.model, "rope_plain_temporal", False))
        N = T * (Hh // 2) * (Ww // 2)
        return (B * N) in self._ws and (T, Hh // 2, Ww // 2, plain) in self._rope

    # -- K3 + K13: conditioning-side work (step-invariant) -------------------------------
    def _context(self, text: torch.Tensor, image: Optional[torch.Tensor]):
        """text [B, Tt, text_dim], image [B, Ti, image_dim] -> per-layer cross-attention K/V, samples stacked along rows."""
        key = None
        if self.model.cache_context:
            # The entry keeps the keyed tensors alive (self._ctx_refs): an address can then not be handed out again for another
            # edit's conditioning while the entry exists, so (data_ptr, _version, shape) identifies the CONTENT, not just a slot.
            key = (text.data_ptr(), text._version, tuple(text.shape), text.dtype,
                   None if image is None else (image.data_ptr(), image._version, tuple(image.shape), image.dtype))
            if key == self._ctx_key:
                return self._ctx
        keyed = (text, image)
        D = self.D
        B, Tt = text.shape[0], text.shape[1]
        text = text.to(torch.bfloat16).reshape(B * Tt, -1)
        if text.shape[1] != self.tx_w1.shape[1]:
            tp = torch.zeros((text.shape[0], self.tx_w1.shape[1]), dtype=torch.bfloat16, device=self.dev)
            tp[:, : text.shape[1]] = text
            text = tp
        t1 = ops.gemm(text.contiguous(), self.tx_w1, self.tx_b1, epilogue=ops.EPI_BIAS_GELU)
        enc_t = ops.gemm(t1, self.tx_w2, self.tx_b2)
        enc_i = None
        Ti = 0
        if image is not None:
            if not self.has_image:
                raise ValueError("encoder_hidden_states_image given but the model has no image_embedder (image_dim=None)")
            Ti = image.shape[1]
            image = image.to(torch.bfloat16).reshape(B * Ti, -1).contiguous()
            w, b, eps = self.im_n1
            h = ops.ln_affine(image, w, b, eps)
            if h.shape[1] != self.im_w1.shape[1]:
                hp = torch.zeros((h.shape[0], self.im_w1.shape[1]), dtype=torch.bfloat16, device=self.dev)
                hp[:, : h.shape[1]] = h
                h = hp
            h = ops.gemm(h, self.im_w1, self.im_b1, epilogue=ops.EPI_BIAS_GELU_ERF)
            if h.shape[1] != self.im_w2.shape[1]:
                hp = torch.zeros((h.shape[0], self.im_w2.shape[1]), dtype=torch.bfloat16, device=self.dev)
                hp[:, : h.shape[1]] = h
                h = hp
            h_img = ops.gemm(h, self.im_w2, self.im_b2)
            w, b, eps = self.im_n2
            enc_i = ops.ln_affine(h_img, w, b, eps)
        # K13: per-layer cross-attention K/V of the text and image context, all layers per launch: K row-major [B*len, L*D] (layer li
        # = columns [li D, (li+1) D)), V either row-major the same way or TRANSPOSED [L*D, columns] (layer li = rows [li D, (li+1) D),
        # sample b's keys at columns [b cols, b cols + len)) for ce_attention_2seg_vt_bf16.  The V^T form needs the N axis of its
        # swapped-role GEMM (= key rows of all samples) in multiples of 8 with even per-sample strides: Tt % 8 == 0 (512), and the image
        # context (257 rows) is layer-normed a second time into a copy whose samples are padded to 264 rows (zero rows -> bias-only
        # columns that meet P = 0).
        eps = self.cfg.eps
        hd = self.cfg.attention_head_dim
        L = self.L
        f8 = self.fp8_cross and (self.model._sp is None or not self.model._sp.sharded) and getattr(self.model, "_cfgp", None) is None
        use_vt = self.cross_vt and enc_i is not None and self.all_img and Tt % 8 == 0 and not f8
        k_t_all = ops.gemm(enc_t, self.w_k_t_all, self.b_k_t_all)  # [B*Tt, L*D]
        k_i_all = v_t_all = v_i_all = v1t = v2t = None
        c1 = c2 = 0
        if enc_i is not None and self.all_img:
            k_i_all = ops.gemm(enc_i, self.w_k_i_all, self.b_k_i_all)
        if use_vt:
            c1, c2 = Tt, (Ti + 7) // 8 * 8
            ld1 = ((B - 1) * c1 + (Tt + 63) // 64 * 64 + 7) // 8 * 8
            ld2 = ((B - 1) * c2 + (Ti + 63) // 64 * 64 + 7) // 8 * 8
            bufs = self._ctx_bufs.get((B, Tt, Ti))
            if bufs is None:  # zero-filled once: the padding rows / columns are never written afterwards.  Engine-owned (not per call):
                # a captured step that computes the projections (cache_context off) replays into the same addresses
                z = lambda *sh: torch.zeros(sh, dtype=torch.bfloat16, device=self.dev)
                bufs = SimpleNamespace(enc_i_pad=z(B * c2, D), v1t=z(L * D, ld1), v2t=z(L * D, ld2))
                self._ctx_bufs = dict(list(self._ctx_bufs.items())[-1:] + [((B, Tt, Ti), bufs)])  # the guided pair and the single sample
            v1t, v2t = bufs.v1t, bufs.v2t
            # these engine-owned buffers are about to be rewritten in place, and a cached context of the same shape holds VIEWS into them
            # (its K tensors are its own): whatever was cached is stale from here on - drop it, so a later hit cannot pair the old K with
            # another conditioning's V^T (a call with cache_context off, key None, would otherwise leave the cached key standing)
            self._ctx_key = self._ctx = self._ctx_refs = None
            w, b, eps_i = self.im_n2
            for bi in range(B):
                ops.ln_affine(h_img[bi * Ti:(bi + 1) * Ti], w, b, eps_i, out=bufs.enc_i_pad[bi * c2: bi * c2 + Ti])
            ops.gemm(self.w_v_t_all, enc_t, self.b_v_t_all, out=v1t[:, : B * Tt], epilogue=ops.EPI_BIAS_ROW)
            ops.gemm(self.w_v_i_all, bufs.enc_i_pad, self.b_v_i_all, out=v2t[:, : B * c2], epilogue=ops.EPI_BIAS_ROW)
        else:
            v_t_all = ops.gemm(enc_t, self.w_v_t_all, self.b_v_t_all)
            if k_i_all is not None:
                v_i_all = ops.gemm(enc_i, self.w_v_i_all, self.b_v_i_all)
        kv = []
        for li, p in enumerate(self.blk if not f8 else ()):
            cols = slice(li * D, (li + 1) * D)
            k_t = k_t_all[:, cols]
            ops.rmsnorm_rope_(k_t, p.nk2, None, hd, eps)
            k_i = None
            if k_i_all is not None:
                k_i = k_i_all[:, cols]
                ops.rmsnorm_rope_(k_i, p.nk_i, None, hd, eps)
            if use_vt:
                kv.append((k_t, v1t[cols], k_i, v2t[cols]))
            else:
                kv.append((k_t, v_t_all[:, cols], k_i, None if v_i_all is None else v_i_all[:, cols]))
        if f8:
            # MXFP8 operands of both segments, per layer: K (RMS-normed, no RoPE) quantised along the head channels, V^T tiles quantised along
            # the keys (ce_rmsnorm_rope_mxfp8 reads the un-normed projection: the bf16 loop above, which norms in place, did not run)
            for li, p in enumerate(self.blk):
                cols = slice(li * D, (li + 1) * D)
                seg_t = ops.rmsnorm_rope_mxfp8(k_t_all[:, cols], p.nk2, None, hd, eps) + ops.v_mxfp8_transpose(v_t_all[:, cols], Tt, B, self.H)
                seg_i = None
                if k_i_all is not None:
                    seg_i = ops.rmsnorm_rope_mxfp8(k_i_all[:, cols], p.nk_i, None, hd, eps) + ops.v_mxfp8_transpose(v_i_all[:, cols], Ti, B, self.H)
                kv.append((seg_t, seg_i))
        ctx = SimpleNamespace(kv=kv, Tt=Tt, Ti=Ti, vt=use_vt, c1=c1, c2=c2, f8=f8)
        if key is not None:
            self._ctx_key, self._ctx, self._ctx_refs = key, ctx, keyed
        return ctx

    def clear_context_cache(self):
        """Drop the cached conditioning-side results (called by the pipeline at the start of every edit)."""
        self._ctx_key = self._ctx = self._ctx_refs = None

    # -- the forward (transformer_chronoedit.py:397-476) ---------------------------------
    def forward(self, hidden: torch.Tensor, timestep: torch.Tensor, text: torch.Tensor, image: Optional[torch.Tensor]):
        """hidden [B,C,T,H,W], timestep [B], text [B,Tt,text_dim], image [B,Ti,image_dim] -> [B,Cout,T,H,W] (bf16).

        The B samples' tokens are stacked along the GEMM M axis (weights stream from HBM once for all of them —
        this is how the two classifier-free-guidance passes of pipeline_chronoedit.py:715-735 are batched);
        attention, RoPE and the AdaLN tables stay per sample."""
        cfg, D, H = self.cfg, self.D, self.H
        B, C, T, Hh, Ww = hidden.shape
        if C != cfg.in_channels:
            raise ValueError(f"expected {cfg.in_channels} input channels, got {C}")
        Hp, Wp = Hh // 2, Ww // 2
        N = T * Hp * Wp
        hd = cfg.attention_head_dim
        eps = cfg.eps
        cs = self._rope_table(T, Hp, Wp)  # raises AssertionError for unsupported frame counts (:205)
        sp = self.model._sp
        if sp is not None and sp.sharded:
            if B != 1 and not self.v_transposed:
                raise ValueError("several samples per sequence-parallel forward need the V^T attention path (enable_transposed_v)")
            if self.fp8_attn and not getattr(self, "_warned_fp8_attn_sp", False):
                import warnings
                warnings.warn("enable_fp8_attention() has no effect while the tokens are sharded (Ulysses / CFG parallel): the exchange "
                              "carries bf16 q / k / v and the self-attention runs the bf16 V^T kernel; see attention_path()")
                self._warned_fp8_attn_sp = True
            align = 64 if B > 1 else 1  # B > 1: blocked receive layout, a key tile must not straddle two source ranks' blocks
            Nl = sp.shard(N, align)[0]  # local (zero-padded) token rows
            key = ("sp", T, Hp, Wp, sp.rank, sp.world, Nl)
            if key not in self._rope:
                self._rope[key] = sp.take_rows(cs, N, align).contiguous()
            cs = self._rope[key]
        else:
            sp = None
            Nl = N
        ws = self._workspace(B * Nl)
        hidden = hidden.to(torch.bfloat16).contiguous()
        # integer timesteps as in the diffusers pipeline; floating-point ones (sibling stacks) keep their fraction
        timestep = timestep.to(device=self.dev, dtype=torch.float32 if timestep.is_floating_point() else torch.int64).contiguous()
        rows = [slice(b * Nl, (b + 1) * Nl) for b in range(B)]

        # K1
        if sp is None:
            for b in range(B):
                ops.patchify(hidden[b], self.kpatch, out=ws.cols[rows[b]])
        else:  # only this rank's token rows (zero rows past the last token: wan_video_new_chronoedit.py:1450-1453)
            for b in range(B):
                ops.patchify(hidden[b], self.kpatch, out=ws.cols[rows[b]], row0=sp.rank * Nl, nrows=Nl)
        ops.gemm(ws.cols, self.w_patch, self.b_patch, out=ws.x)

        # K2 per sample: sinusoid -> time_embedder (fp32) -> temb (bf16-rounded) -> silu -> time_proj -> AdaLN tables
        mods, gates1, gates2, mods_out = [], [], [], []
        for b in range(B):
            sin = ops.timestep_sinusoid(timestep[b : b + 1], cfg.freq_dim)
            h1 = ops.gemv(self.te_w1, sin, self.te_b1, flags=2)
            temb = ops.gemv(self.te_w2, h1, self.te_b2, flags=4)
            tproj = ops.gemv(self.tp_w, temb, self.tp_b, flags=1 | 4)  # [6*D]
            mods.append(ops.modulation(self.tables, tproj.view(6, D), one_mask=0b010010))  # [L,6,D]: shift,1+scale,gate,...
            mods_out.append(ops.modulation(self.table_out, temb.view(1, D), one_mask=0b10))  # [1,2,D]: shift, 1+scale
        mod = torch.stack(mods, dim=1).contiguous()  # [L, B, 6, D]: per-sample AdaLN rows (ab_rows / gate_rows = tokens per sample)
        mod_out = torch.stack(mods_out, dim=0).contiguous()  # [B, 1, 2, D]
        if B > 1:  # per-sample gate vectors, stacked [L, B, D] for the GEMM epilogue
            gate_msa = mod[:, :, 2].contiguous()
            gate_ffn = mod[:, :, 5].contiguous()

        if text.shape[0] != B or (image is not None and image.shape[0] != B):
            raise ValueError("encoder_hidden_states / encoder_hidden_states_image batch size must match hidden_states")
        ctx = self._context(text, image)
        Tt, Ti = ctx.Tt, ctx.Ti
        grow = Nl if B > 1 else 0

        x = ws.x
        fuse_o = self.mx and self.fuse_attn_quant and D % 128 == 0  # MX: both attention kernels emit the out-projections' fp8 operands themselves
        fuse_o1, fuse_o2 = fuse_o and "o1" in self.fp8_set, fuse_o and "o2" in self.fp8_set  # (only for an out-projection that runs in fp8)
        for li, p in enumerate(self.blk):
            q_o1 = False
            # 1. self-attention
            if sp is None and self.fp8_attn:  # MXFP8: the norm / RoPE pass and a V^T pass write the quantised operands
                self._ln_linear(ws, x, mod[li, 0, 1], mod[li, 0, 0], p, "qkv", ws.qkv, ab_rows=Nl, ab_stride=6 * D)
                ops.rmsnorm_rope_mxfp8(ws.qkv[:, :D], p.nq1, cs, hd, eps, out=ws.q8, scale=ws.sq, post_scale=ops.MXFP8_Q_SCALE)
                ops.rmsnorm_rope_mxfp8(ws.qkv[:, D : 2 * D], p.nk1, cs, hd, eps, out=ws.k8, scale=ws.sk)
                if ws.v8t is None or ws.v8t.shape[0] != B:
                    ws.v8t = ws.sv = None
                ws.v8t, ws.sv = ops.v_mxfp8_transpose(ws.qkv[:, 2 * D :], Nl, B, H, out=ws.v8t, scale=ws.sv)
                if fuse_o1:  # the out-projection's MX operand straight from the attention epilogue (no bf16 output, no quantisation pass)
                    ops.attention_mxfp8(ws.q8, ws.sq, ws.k8, ws.sk, ws.v8t, ws.sv, H, batch=B, out8=ws.a8[:, :D], scale8=ws.s8)
                    q_o1 = True
                else:
                    ops.attention_mxfp8(ws.q8, ws.sq, ws.k8, ws.sk, ws.v8t, ws.sv, H, out=ws.att, batch=B)
                att = ws.att
            elif sp is None and self.v_transposed and "qkv" not in self.fp8_set and (B * Nl) % 8 == 0 and (B == 1 or Nl % 2 == 0):
                # q | k as one GEMM, V^T = (h.W_v^T)^T by a GEMM whose epilogue stores the transpose (round 6; rounds 2-5: the operand roles
                # swapped, bias along rows - what shapes outside the large tile still run): the attention kernel's V^T operand [D][keys of all
                # samples] without a transpose pass; K and V^T tiles both go by LDS-DMA
                if getattr(ws, "vt", None) is None:
                    ws.vt = torch.zeros((D, ops.vt_columns(B * Nl)), dtype=torch.bfloat16, device=self.dev)  # padding columns stay zero
                ops.ln_affine(x, mod[li, 0, 1], mod[li, 0, 0], eps, out=ws.h, ab_rows=Nl, ab_stride=6 * D)
                ops.gemm(ws.h, p.w_qkv[: 2 * D], p.b_qkv[: 2 * D], out=ws.qkv[:, : 2 * D])
                if _VT_GEMM_SWAPPED:  # (A/B knob of tools: CE_VT_GEMM=row - the rounds-2-5 form of the same product)
                    ops.gemm(p.w_qkv[2 * D :], ws.h, p.b_qkv[2 * D :], out=ws.vt[:, : B * Nl], epilogue=ops.EPI_BIAS_ROW)
                else:
                    ops.gemm(ws.h, p.w_qkv[2 * D :], p.b_qkv[2 * D :], out=ws.vt[:, : B * Nl], epilogue=ops.EPI_BIAS_T)  # (stores the transpose: V^T)
                ops.rmsnorm_rope_(ws.qkv[:, :D], p.nq1, cs, hd, eps, x2=ws.qkv[:, D : 2 * D], w2=p.nk1)
                ops.attention_vt(ws.qkv[:, :D], ws.qkv[:, D : 2 * D], ws.vt, H, out=ws.att, batch=B)
                att = ws.att
            elif sp is None:  # all samples in one launch (stacked rows)
                self._ln_linear(ws, x, mod[li, 0, 1], mod[li, 0, 0], p, "qkv", ws.qkv, ab_rows=Nl, ab_stride=6 * D)
                ops.rmsnorm_rope_(ws.qkv[:, :D], p.nq1, cs, hd, eps, x2=ws.qkv[:, D : 2 * D], w2=p.nk1)  # q and k, all samples
                ops.attention(ws.qkv[:, :D], ws.qkv[:, D : 2 * D], ws.qkv[:, 2 * D :], H, out=ws.att, batch=B)
                att = ws.att
            else:
                att = self._self_attention_ulysses(ws, sp, x, mod[li, 0, 1], mod[li, 0, 0], p, cs, N, Nl, B)
            self._linear(ws, att, p, "o1", x, quantised=q_o1, epilogue=ops.EPI_GATE_RES, gate=gate_msa[li] if B > 1 else mods[0][li, 2],
                         res=x, gate_rows=grow)
            # 2. cross-attention (text + image segments)
            if p.n2w is not None:
                self._ln_linear(ws, x, p.n2w, p.n2b, p, "q2", ws.q2)
            else:
                self._linear(ws, x, p, "q2", ws.q2)
            q_o2 = bool(ctx.vt and fuse_o2)
            if ctx.f8:  # MXFP8 (round 5): text segment -> bf16, image segment adds it and emits bf16 or the out-projection's MX operand
                seg_t, seg_i = ctx.kv[li]
                ops.rmsnorm_rope_mxfp8(ws.q2, p.nq2, None, hd, eps, out=ws.q8, scale=ws.sq, post_scale=ops.MXFP8_Q_SCALE)
                if seg_i is None and fuse_o2:
                    ops.attention_mxfp8(ws.q8, ws.sq, *seg_t, H, batch=B, out8=ws.a8[:, :D], scale8=ws.s8)
                else:
                    ops.attention_mxfp8(ws.q8, ws.sq, *seg_t, H, out=ws.att, batch=B)
                    if seg_i is not None and fuse_o2:
                        ops.attention_mxfp8(ws.q8, ws.sq, *seg_i, H, batch=B, out8=ws.a8[:, :D], scale8=ws.s8, add=ws.att)
                    elif seg_i is not None:
                        ops.attention_mxfp8(ws.q8, ws.sq, *seg_i, H, out=ws.att, batch=B, add=ws.att)
                q_o2 = fuse_o2
                self._linear(ws, ws.att, p, "o2", x, quantised=q_o2, epilogue=ops.EPI_GATE_RES, gate=None, res=x)
                k_t = None
            else:
                ops.rmsnorm_rope_(ws.q2, p.nq2, None, hd, eps)
                k_t, v_t, k_i, v_i = ctx.kv[li]
            if k_t is None:
                pass
            elif ctx.vt and fuse_o2:
                ops.attention_2seg_vt(ws.q2, k_t, v_t, Tt, k_i, v_i, Ti, H, batch=B, cols1=ctx.c1, cols2=ctx.c2, out8=ws.a8[:, :D], scale8=ws.s8)
            elif ctx.vt:  # both segments' K and V^T tiles by LDS-DMA (v_t / v_i are V^T row blocks of this layer)
                ops.attention_2seg_vt(ws.q2, k_t, v_t, Tt, k_i, v_i, Ti, H, out=ws.att, batch=B, cols1=ctx.c1, cols2=ctx.c2)
            elif k_i is not None:
                ops.attention(ws.q2, k_t, v_t, H, out=ws.att, k2=k_i, v2=v_i, batch=B)
            else:
                ops.attention(ws.q2, k_t, v_t, H, out=ws.att, batch=B)
            if k_t is not None:
                self._linear(ws, ws.att, p, "o2", x, quantised=q_o2, epilogue=ops.EPI_GATE_RES, gate=None, res=x)
            # 3. feed-forward
            if self.mx and self.fuse_quant and "f1" in self.fp8_set and "f2" in self.fp8_set:
                # MX: the up-projection's bias + GELU epilogue emits the down-projection's fp8 operand and its block scales directly (a
                # block = 32 consecutive output columns: no row-wide reduction) - no bf16 [N, F] matrix, no quantisation pass
                aq = ws.a8[:, :D]
                ops.ln_affine_mxfp8(x, mod[li, 0, 4], mod[li, 0, 3], eps, out=aq, scale=ws.s8, ab_rows=Nl, ab_stride=6 * D)
                ops.gemm_mxfp8_gelu_quant(aq, ws.s8, *p.q_f1, p.b_f1, out=ws.a8b, scale=ws.s8b)
                ops.gemm_mxfp8(ws.a8b, ws.s8b, *p.q_f2, p.b_f2, out=x, epilogue=ops.EPI_GATE_RES,
                               gate=gate_ffn[li] if B > 1 else mods[0][li, 5], res=x, gate_rows=grow)
            else:
                self._ln_linear(ws, x, mod[li, 0, 4], mod[li, 0, 3], p, "f1", ws.ffn, ab_rows=Nl, ab_stride=6 * D, epilogue=ops.EPI_BIAS_GELU)
                self._linear(ws, ws.ffn, p, "f2", x, epilogue=ops.EPI_GATE_RES, gate=gate_ffn[li] if B > 1 else mods[0][li, 5],
                             res=x, gate_rows=grow)

        # K18
        ops.ln_affine(x, mod_out[0, 0, 1], mod_out[0, 0, 0], eps, out=ws.h, ab_rows=Nl, ab_stride=2 * D)
        ops.gemm(ws.h, self.w_out, self.b_out, out=ws.head)
        out = torch.empty((B, cfg.out_channels, T, Hh, Ww), dtype=torch.bfloat16, device=self.dev)
        if sp is None:
            for b in range(B):
                ops.unpatchify(ws.head[rows[b]], cfg.out_channels, T, Hh, Ww, out=out[b])
        else:
            full = sp.all_gather_rows(ws.head).view(sp.world, B, Nl, -1)  # [source rank][sample][local token]
            for b in range(B):
                ops.unpatchify(full[:, b].reshape(sp.world * Nl, -1)[:N].contiguous(), cfg.out_channels, T, Hh, Ww, out=out[b])
        return out
