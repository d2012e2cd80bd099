"""CPU oracle for the ChronoEdit DiT forward (TEST INFRASTRUCTURE — never the product path).

A plain-PyTorch, CPU-only restatement of the reference's denoiser

    /root/reference/chronoedit_diffusers/transformer_chronoedit.py:38-476

written as free functions over a *state dict with the diffusers key names*
(/root/reference/chronoedit_diffsynth/wan_video_dit_chronoedit.py:439-496 lists
them).  The pieces the reference delegates to the un-vendored ``diffusers==0.35.2``
wheel are restated from their published behaviour and cross-checked against the two
in-repo sibling renderings (citations at each function).

Pinning status: the reference ships no golden vectors for this path.
``oracle/gen_golden.py`` executes the reference's *own* transformer_chronoedit.py in
this container (with a stand-in for the absent diffusers leaf modules) and commits
the outputs under tests/golden/; tests/test_oracle_golden.py checks this file
against them.  The diffusers leaf-module semantics themselves remain unpinned
("parity unpinned" for those leaves).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

dtype islands (bf16 run) follow SURVEY.md Appendix A / transformer_chronoedit.py:274-293:
every ``.float()`` / ``.type_as`` of the reference is reproduced, so running this
oracle with bf16 parameters reproduces the reference's eager bf16 rounding points.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class DiTConfig:
    """Constructor arguments of ChronoEditTransformer3DModel
    (transformer_chronoedit.py:341-360); 14B values from
    wan_video_dit_chronoedit.py:525-538."""

    patch_size: Tuple[int, int, int] = (1, 2, 2)
    num_attention_heads: int = 40
    attention_head_dim: int = 128
    in_channels: int = 36
    out_channels: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    ffn_dim: int = 13824
    num_layers: int = 40
    cross_attn_norm: bool = True
    qk_norm: str = "rms_norm_across_heads"
    eps: float = 1e-6
    image_dim: Optional[int] = 1280
    added_kv_proj_dim: Optional[int] = 5120
    rope_max_seq_len: int = 1024
    rope_temporal_skip_len: int = 8
    rope_plain_temporal: bool = False  # temporal positions 0..T-1 for any T: the diffsynth call path (wan_video_new_chronoedit.py:1428-1432)

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


# --------------------------------------------------------------------------------------
# parameter construction (SURVEY.md §8d "Synthetic inputs")
# --------------------------------------------------------------------------------------

FP32_KEEP = ("time_embedder", "scale_shift_table", "norm1", "norm2", "norm3")
"""_keep_in_fp32_modules, transformer_chronoedit.py:338"""


def param_shapes(cfg: DiTConfig) -> Dict[str, Tuple[int, ...]]:
    """Parameter tree of ChronoEditTransformer3DModel with diffusers names
    (transformer_chronoedit.py:366-393; key list wan_video_dit_chronoedit.py:439-496)."""
    D, Fd = cfg.inner_dim, cfg.ffn_dim
    pt, ph, pw = cfg.patch_size
    s: Dict[str, Tuple[int, ...]] = {}
    s["patch_embedding.weight"] = (D, cfg.in_channels, pt, ph, pw)
    s["patch_embedding.bias"] = (D,)
    ce = "condition_embedder."
    s[ce + "time_embedder.linear_1.weight"] = (D, cfg.freq_dim)
    s[ce + "time_embedder.linear_1.bias"] = (D,)
    s[ce + "time_embedder.linear_2.weight"] = (D, D)
    s[ce + "time_embedder.linear_2.bias"] = (D,)
    s[ce + "time_proj.weight"] = (6 * D, D)
    s[ce + "time_proj.bias"] = (6 * D,)
    s[ce + "text_embedder.linear_1.weight"] = (D, cfg.text_dim)
    s[ce + "text_embedder.linear_1.bias"] = (D,)
    s[ce + "text_embedder.linear_2.weight"] = (D, D)
    s[ce + "text_embedder.linear_2.bias"] = (D,)
    if cfg.image_dim is not None:
        I = cfg.image_dim
        s[ce + "image_embedder.norm1.weight"] = (I,)
        s[ce + "image_embedder.norm1.bias"] = (I,)
        s[ce + "image_embedder.ff.net.0.proj.weight"] = (I, I)
        s[ce + "image_embedder.ff.net.0.proj.bias"] = (I,)
        s[ce + "image_embedder.ff.net.2.weight"] = (D, I)
        s[ce + "image_embedder.ff.net.2.bias"] = (D,)
        s[ce + "image_embedder.norm2.weight"] = (D,)
        s[ce + "image_embedder.norm2.bias"] = (D,)
    for i in range(cfg.num_layers):
        b = f"blocks.{i}."
        for a in ("attn1", "attn2"):
            for p in ("to_q", "to_k", "to_v", "to_out.0"):
                s[b + f"{a}.{p}.weight"] = (D, D)
                s[b + f"{a}.{p}.bias"] = (D,)
            s[b + f"{a}.norm_q.weight"] = (D,)
            s[b + f"{a}.norm_k.weight"] = (D,)
        if cfg.added_kv_proj_dim is not None:
            s[b + "attn2.add_k_proj.weight"] = (D, cfg.added_kv_proj_dim)
            s[b + "attn2.add_k_proj.bias"] = (D,)
            s[b + "attn2.add_v_proj.weight"] = (D, cfg.added_kv_proj_dim)
            s[b + "attn2.add_v_proj.bias"] = (D,)
            s[b + "attn2.norm_added_k.weight"] = (D,)
        if cfg.cross_attn_norm:
            s[b + "norm2.weight"] = (D,)
            s[b + "norm2.bias"] = (D,)
        s[b + "ffn.net.0.proj.weight"] = (Fd, D)
        s[b + "ffn.net.0.proj.bias"] = (Fd,)
        s[b + "ffn.net.2.weight"] = (D, Fd)
        s[b + "ffn.net.2.bias"] = (D,)
        s[b + "scale_shift_table"] = (1, 6, D)
    s["proj_out.weight"] = (cfg.out_channels * pt * ph * pw, D)
    s["proj_out.bias"] = (cfg.out_channels * pt * ph * pw,)
    s["scale_shift_table"] = (1, 2, D)
    return s


def keep_fp32(name: str) -> bool:
    return any(k in name for k in FP32_KEEP)


def make_synthetic_params(
    cfg: DiTConfig, seed: int = 1234, dtype: torch.dtype = torch.float32, bias_std: float = 0.02
) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: Linear/Conv ~ N(0, 0.02^2); norm weights
    1 + N(0, 0.02^2) (so a dropped affine is visible); biases N(0, bias_std^2);
    scale_shift_table ~ N(0,1)/sqrt(D) as in the ctor (transformer_chronoedit.py:265,393).
    Under bf16, the _keep_in_fp32_modules stay fp32 (transformer_chronoedit.py:338)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    D = cfg.inner_dim
    for name, shape in param_shapes(cfg).items():
        if name.endswith("scale_shift_table"):
            t = torch.randn(shape, generator=g) / D**0.5
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.02 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = bias_std * torch.randn(shape, generator=g)
        else:
            t = 0.02 * torch.randn(shape, generator=g)
        out[name] = t.to(torch.float32 if keep_fp32(name) else dtype)
    return out


# --------------------------------------------------------------------------------------
# leaf semantics that live in diffusers 0.35.2 (SURVEY.md §8c list)
# --------------------------------------------------------------------------------------


def fp32_layer_norm(x, weight, bias, eps):
    """diffusers FP32LayerNorm: layer_norm in fp32, cast back to the input dtype.
    Sibling: chronoedit/_src/networks/wan2pt1.py:256-266."""
    return F.layer_norm(
        x.float(),
        (x.shape[-1],),
        weight.float() if weight is not None else None,
        bias.float() if bias is not None else None,
        eps,
    ).to(x.dtype)


def rms_norm(x, weight, eps):
    """diffusers RMSNorm (qk_norm="rms_norm_across_heads" -> RMSNorm(heads*dim_head)):
    variance in fp32, x*rsqrt in fp32, cast to the weight dtype when that is half
    precision, then multiply.  Sibling: wan_video_dit_chronoedit.py:115-126."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        y = y.to(weight.dtype)
    return y * weight


def timestep_sinusoid(timestep: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0):
    fp32 [cos, sin] of t * 10000^(-i/half).  Siblings compute the same table in fp64
    (wan_video_dit_chronoedit.py:83-87, wan2pt1.py:191-200)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timestep[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def rope_freqs_1d(dim: int, max_len: int, theta: float = 10000.0) -> torch.Tensor:
    """diffusers get_1d_rotary_pos_embed(dim, max_len, theta, use_real=False,
    freqs_dtype=float64) -> complex128 [max_len, dim/2].
    Sibling: wan_video_dit_chronoedit.py:98-104."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
    ang = torch.outer(torch.arange(max_len, dtype=torch.float64), freqs)
    return torch.polar(torch.ones_like(ang), ang)


def rope_table(cfg: DiTConfig, num_frames: int, height: int, width: int) -> torch.Tensor:
    """ChronoEditRotaryPosEmbed (transformer_chronoedit.py:168-213).  Returns complex128
    [1, 1, N, head_dim/2].  With 2 latent frames the temporal indices are {0, skip_len-1}
    (transformer_chronoedit.py:205-207)."""
    hd = cfg.attention_head_dim
    h_dim = w_dim = 2 * (hd // 6)
    t_dim = hd - h_dim - w_dim
    ft, fh, fw = (rope_freqs_1d(d, cfg.rope_max_seq_len) for d in (t_dim, h_dim, w_dim))
    pt, ph, pw = cfg.patch_size
    ppf, pph, ppw = num_frames // pt, height // ph, width // pw
    assert cfg.rope_plain_temporal or num_frames == 2 or num_frames == cfg.rope_temporal_skip_len, (
        f"num_frames must be 2 or {cfg.rope_temporal_skip_len}, but got {num_frames}"
    )
    if num_frames == 2 and not cfg.rope_plain_temporal:
        f_t = ft[: cfg.rope_temporal_skip_len][[0, -1]]
    else:
        f_t = ft[:ppf]
    f_t = f_t.view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1)
    f_h = fh[:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1)
    f_w = fw[:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)
    return torch.cat([f_t, f_h, f_w], dim=-1).reshape(1, 1, ppf * pph * ppw, -1)


def apply_rope(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """transformer_chronoedit.py:73-76: pairs viewed as complex128, multiplied, cast back."""
    xc = torch.view_as_complex(x.to(torch.float64).unflatten(3, (-1, 2)))
    return torch.view_as_real(xc * freqs).flatten(3, 4).type_as(x)


def linear(x, p, name):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def fake_quant_rows_fp8(x: torch.Tensor) -> torch.Tensor:
    """The fp8 contract of chronoedit_amd (include/chronoedit_hip.h, ce_quant_rows_fp8): per row s = amax / 448 (1 for a zero row),
    values rounded to OCP e4m3 (torch.float8_e4m3fn, round to nearest even), returned de-quantised in x's dtype."""
    amax = x.float().abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return ((x.float() / s).to(torch.float8_e4m3fn).float() * s).to(x.dtype)


def linear_fp8(x, p, name):
    """A Linear under the fp8 contract: activations quantised per token row, weights per output channel, exact products, bias
    added afterwards (BASELINE.json configs[4]; the reference has no fp8 path of its own)."""
    return F.linear(fake_quant_rows_fp8(x), fake_quant_rows_fp8(p[name + ".weight"]), p[name + ".bias"])


def linear_mxfp8(x, p, name):
    """A Linear under the MX form of the fp8 contract (round 4: include/chronoedit_hip.h ce_gemm_mxfp8): activations and weights as OCP
    MXFP8 - e4m3 elements, one E8M0 scale per 32 consecutive INPUT channels of a row (mx_quant below, non-saturating scale choice: the
    smallest power of two with amax / scale <= 448) - exact products, block scales applied to the partial sums, bias added afterwards.  (The reference has no fp8 path: this is the definition.)"""
    return F.linear(mx_quant(x, -1, saturating=False).to(x.dtype), mx_quant(p[name + ".weight"], -1, saturating=False).to(x.dtype), p[name + ".bias"])


def _fp8_linear(fp8):
    """fp8 = False: plain Linear; True / "row": per-row scales (linear_fp8); "mx": MX block scales (linear_mxfp8)."""
    if not fp8:
        return linear
    return linear_mxfp8 if fp8 == "mx" else linear_fp8


def mx_quant(x: torch.Tensor, dim: int = -1, block_index: Optional[torch.Tensor] = None, saturating: bool = True) -> torch.Tensor:
    """OCP MXFP8 (e4m3 elements, one E8M0 scale per 32 elements along `dim`), returned de-quantised in fp32: the contract of
    chronoedit_amd/csrc/ce_attn_fp8.hip.  scale = 2^(floor(log2 amax) - 8) (2^-126 for an all-zero block), elements =
    RNE(x / scale) clamped to +-448 (saturating=True: the OCP floor rule, what the attention operands use; saturating=False: one step up
    when amax / scale would exceed 448, what the GEMM operands use).  block_index (optional, [size of dim] long): block id of every element along `dim` when the
    32-element blocks are not the contiguous ones (the V operand: see attention_mxfp8)."""
    xf = x.float().movedim(dim, -1)
    n = xf.shape[-1]
    if block_index is None and n % 32 == 0:  # contiguous whole blocks: the same arithmetic without the scatter (weights of 70 M elements)
        xb = xf.reshape(xf.shape[:-1] + (n // 32, 32))
        amax = xb.abs().amax(-1, keepdim=True)
        e = torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -118))) - 8.0
        if not saturating:  # the smallest power of two with amax / scale <= 448 (the GEMM operands: ce_common.h mx_scale_byte_nosat)
            e = e + (amax > 448.0 * torch.exp2(e)).float()
        e = torch.where(amax > 0, torch.clamp(e, min=-126.0), torch.full_like(e, -126.0))
        scale = torch.exp2(e)
        q = torch.clamp(xb / scale, -448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
        return q.reshape(xf.shape).movedim(-1, dim)
    if block_index is None:
        block_index = torch.arange(n) // 32
    nb = int(block_index.max()) + 1
    amax = torch.zeros(xf.shape[:-1] + (nb,), dtype=torch.float32)
    amax = amax.index_reduce(-1, block_index, xf.abs(), "amax", include_self=True)
    e = torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -118))) - 8.0
    e = torch.where(amax > 0, torch.clamp(e, min=-126.0), torch.full_like(e, -126.0))
    scale = torch.exp2(e).index_select(-1, block_index)
    q = torch.clamp(xf / scale, -448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    return q.movedim(-1, dim)


def attention_mxfp8(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, lazy_offset: bool = True) -> torch.Tensor:
    """[B, H, N, 128] q / k (normalised, rotated) and v -> softmax(q k^T / sqrt(128)) v under the MXFP8 contract of ce_attn_fp8.hip:
    q * (softmax_scale * log2 e) and k quantised in blocks of 32 consecutive head channels, v in blocks of 32 consecutive KEYS,
    exact fp32 products (scores in the exp2 domain); online softmax over 64-key tiles: P = exp2(S - offset) rounded to e4m3 (unit
    scale), O and l rescaled by exp2(old - new offset) when the offset moves; O / l at the end.  Two offset schedules:
      lazy_offset=True   (default kernel, variant 1) the offset of a row is an INTEGER: ceil(row maximum of tile 0 - 3), so P <= 8;
                         afterwards it is raised - for the 32 query rows one wave owns together - only when an element of a tile
                         converts to NaN (exp2(S - offset) > 464, above the e4m3 range): then each of those rows takes
                         max(offset, ceil(its tile maximum - 3)).  l is the sum of the ROUNDED P (it leaves the matrix pipe).
      lazy_offset=False  (plain loop, variant 0) running row maximum - 3 after every tile; l is the fp32 sum of the un-rounded P.
    e4m3 is a floating-point format, so the two schedules differ in individual roundings of P, not in their distribution.
    The reference has no fp8 path (transformer_chronoedit.py:91-104 is plain SDPA): this IS the definition."""
    B, H, N, hd = q.shape
    Nk = k.shape[2]
    c = hd ** -0.5 * 1.4426950408889634
    qq, kq = mx_quant(q.float() * c, -1), mx_quant(k, -1)
    vq = mx_quant(v, 2)
    off = torch.full((B, H, N, 1), -1.0e30)
    l = torch.zeros((B, H, N, 1))
    o = torch.zeros((B, H, N, hd))
    nblk = (N + 31) // 32
    for ti, t0 in enumerate(range(0, Nk, 64)):
        s = qq @ kq[:, :, t0:t0 + 64].transpose(-1, -2)
        tmax = s.amax(-1, keepdim=True)
        if not lazy_offset:
            new_off = torch.maximum(off, tmax - 3.0)
        elif ti == 0:
            new_off = torch.ceil(tmax - 3.0)
        else:
            over = (torch.exp2(s - off) > 464.0).any(-1)                                   # [B, H, N]: an element would convert to NaN
            over = torch.nn.functional.pad(over, (0, nblk * 32 - N)).view(B, H, nblk, 32)
            trig = over.any(-1, keepdim=True).expand(B, H, nblk, 32).reshape(B, H, nblk * 32)[..., :N, None]
            new_off = torch.where(trig, torch.maximum(off, torch.ceil(tmax - 3.0)), off)
        alpha = torch.exp2(off - new_off)
        pr = torch.exp2(s - new_off)
        p8 = pr.to(torch.float8_e4m3fn).float()
        o = o * alpha + p8 @ vq[:, :, t0:t0 + 64]
        l = l * alpha + (p8 if lazy_offset else pr).sum(-1, keepdim=True)
        off = new_off
    return (o / l).to(q.dtype)


# --------------------------------------------------------------------------------------
# the modules of transformer_chronoedit.py
# --------------------------------------------------------------------------------------


FP8_LINEARS = ("qkv", "o1", "q2", "o2", "f1", "f2")  # the names chronoedit_amd.transformer gives the six large Linears of a block


def attention(p, pre, cfg: DiTConfig, hidden, encoder=None, rotary=None, taps=None, fp8=False, fp8_attn=False, fp8_linears=FP8_LINEARS):
    """ChronoEditAttnProcessor2_0.__call__ (transformer_chronoedit.py:43-108).  fp8: the projections that chronoedit_amd runs on
    the fp8 path (q, the self-attention k / v, the output projection) follow linear_fp8; the context k / v stay as they are.
    fp8_attn: True - the SELF-attention product under the MXFP8 contract (attention_mxfp8), cross-attention stays SDPA; "all" (round 5) -
    the two segments of the cross-attention under the same contract too, each rounded to the activation dtype before the add (:96-107)."""
    self_attn = encoder is None
    H = cfg.num_attention_heads
    # mixed precision (round 6): fp8_linears names which of the Linears follow the fp8 contract - self-attention: "qkv" (to_q / to_k / to_v) and
    # "o1" (to_out); cross-attention: "q2" (to_q) and "o2" (to_out); the rest are plain
    lin_q = _fp8_linear(fp8 if ("qkv" if self_attn else "q2") in fp8_linears else False)
    lin_kv = _fp8_linear(fp8 if "qkv" in fp8_linears else False) if encoder is None else linear
    lin_o = _fp8_linear(fp8 if ("o1" if self_attn else "o2") in fp8_linears else False)
    enc_img = None
    has_added = (pre + ".add_k_proj.weight") in p
    if has_added and encoder is not None:
        enc_img, encoder = encoder[:, :257], encoder[:, 257:]
    if encoder is None:
        encoder = hidden
    q = lin_q(hidden, p, pre + ".to_q")
    k = lin_kv(encoder, p, pre + ".to_k")
    v = lin_kv(encoder, p, pre + ".to_v")
    q = rms_norm(q, p[pre + ".norm_q.weight"], cfg.eps)
    k = rms_norm(k, p[pre + ".norm_k.weight"], cfg.eps)
    q = q.unflatten(2, (H, -1)).transpose(1, 2)
    k = k.unflatten(2, (H, -1)).transpose(1, 2)
    v = v.unflatten(2, (H, -1)).transpose(1, 2)
    if rotary is not None:
        q = apply_rope(q, rotary)
        k = apply_rope(k, rotary)
    if taps is not None:
        taps[pre + ".q"], taps[pre + ".k"], taps[pre + ".v"] = q, k, v
    out_img = None
    if enc_img is not None:
        k_img = linear(enc_img, p, pre + ".add_k_proj")
        k_img = rms_norm(k_img, p[pre + ".norm_added_k.weight"], cfg.eps)
        v_img = linear(enc_img, p, pre + ".add_v_proj")
        k_img = k_img.unflatten(2, (H, -1)).transpose(1, 2)
        v_img = v_img.unflatten(2, (H, -1)).transpose(1, 2)
        out_img = attention_mxfp8(q, k_img, v_img) if fp8_attn == "all" else F.scaled_dot_product_attention(q, k_img, v_img)
        out_img = out_img.transpose(1, 2).flatten(2, 3).type_as(q)
    out = attention_mxfp8(q, k, v) if (fp8_attn and (self_attn or fp8_attn == "all")) else F.scaled_dot_product_attention(q, k, v)
    out = out.transpose(1, 2).flatten(2, 3).type_as(q)
    if out_img is not None:
        out = out + out_img
    if taps is not None:
        taps[pre + ".sdpa"] = out
    return lin_o(out, p, pre + ".to_out.0")


def feed_forward(p, pre, x, approximate: str, fp8=False, fp8_linears=FP8_LINEARS):
    """diffusers FeedForward: net.0 = GELU(proj + gelu), net.1 = Dropout(0), net.2 = Linear.
    "gelu-approximate" -> tanh (block FFN, transformer_chronoedit.py:262);
    "gelu" -> erf (image MLP, :116).  Sibling: wan_video_dit_chronoedit.py:224-225,251-257."""
    lin1, lin2 = _fp8_linear(fp8 if "f1" in fp8_linears else False), _fp8_linear(fp8 if "f2" in fp8_linears else False)
    h = F.gelu(lin1(x, p, pre + ".net.0.proj"), approximate=approximate)
    return lin2(h, p, pre + ".net.2")


def block_forward(p, i: int, cfg: DiTConfig, x, encoder, temb6, rotary, taps=None, fp8=False, fp8_attn=False, fp8_linears=FP8_LINEARS):
    """ChronoEditTransformerBlock.forward (transformer_chronoedit.py:267-295)."""
    b = f"blocks.{i}"
    shift, scale, gate, c_shift, c_scale, c_gate = (p[b + ".scale_shift_table"] + temb6.float()).chunk(6, dim=1)
    h = (fp32_layer_norm(x.float(), None, None, cfg.eps) * (1 + scale) + shift).type_as(x)
    if taps is not None:
        taps[b + ".ln1"] = h
    a = attention(p, b + ".attn1", cfg, h, None, rotary, taps, fp8=fp8, fp8_attn=fp8_attn, fp8_linears=fp8_linears)
    x = (x.float() + a * gate).type_as(x)
    if taps is not None:
        taps[b + ".x_after_attn1"] = x
    if cfg.cross_attn_norm:
        h = fp32_layer_norm(x.float(), p[b + ".norm2.weight"], p[b + ".norm2.bias"], cfg.eps).type_as(x)
    else:
        h = x
    a = attention(p, b + ".attn2", cfg, h, encoder, None, taps, fp8=fp8, fp8_linears=fp8_linears)
    x = x + a
    if taps is not None:
        taps[b + ".x_after_attn2"] = x
    h = (fp32_layer_norm(x.float(), None, None, cfg.eps) * (1 + c_scale) + c_shift).type_as(x)
    f = feed_forward(p, b + ".ffn", h, "tanh", fp8=fp8, fp8_linears=fp8_linears)
    x = (x.float() + f.float() * c_gate).type_as(x)
    return x


def condition_embed(p, cfg: DiTConfig, timestep, text, image):
    """ChronoEditTimeTextImageEmbedding.forward (transformer_chronoedit.py:147-165) and
    ChronoEditImageEmbedding.forward (:119-123)."""
    ce = "condition_embedder."
    t = timestep_sinusoid(timestep, cfg.freq_dim)
    te_dtype = p[ce + "time_embedder.linear_1.weight"].dtype
    t = t.to(te_dtype)
    temb = linear(F.silu(linear(t, p, ce + "time_embedder.linear_1")), p, ce + "time_embedder.linear_2")
    temb = temb.type_as(text)
    timestep_proj = linear(F.silu(temb), p, ce + "time_proj")
    # PixArtAlphaTextProjection(act_fn="gelu_tanh")
    text = linear(F.gelu(linear(text, p, ce + "text_embedder.linear_1"), approximate="tanh"), p, ce + "text_embedder.linear_2")
    if image is not None:
        ie = ce + "image_embedder."
        h = fp32_layer_norm(image, p[ie + "norm1.weight"], p[ie + "norm1.bias"], 1e-5)
        h = feed_forward(p, ie + "ff", h, "none")
        image = fp32_layer_norm(h, p[ie + "norm2.weight"], p[ie + "norm2.bias"], 1e-5)
    return temb, timestep_proj, text, image


def dit_forward(
    p: Dict[str, torch.Tensor],
    cfg: DiTConfig,
    hidden_states: torch.Tensor,
    timestep: torch.Tensor,
    encoder_hidden_states: torch.Tensor,
    encoder_hidden_states_image: Optional[torch.Tensor] = None,
    taps: Optional[dict] = None,
    fp8=False,
    fp8_attn: bool = False,
    fp8_linears=FP8_LINEARS,
) -> torch.Tensor:
    """ChronoEditTransformer3DModel.forward (transformer_chronoedit.py:397-476).  fp8 restates chronoedit_amd's fp8 GEMM modes (the six
    large Linears of every block under linear_fp8 - fp8=True / "row": per-row scales - or linear_mxfp8 - fp8="mx": MX block scales; everything
    else unchanged); fp8_attn=True its MXFP8 self-attention, fp8_attn="all" the cross-attention under that contract as well; fp8_linears: the
    subset of the six that follow the fp8 contract (the engine's mixed-precision policies, ChronoEditTransformer3DModel.FP8_POLICIES)."""
    B, C, T, Hh, Ww = hidden_states.shape
    pt, ph, pw = cfg.patch_size
    ppf, pph, ppw = T // pt, Hh // ph, Ww // pw
    rotary = rope_table(cfg, T, Hh, Ww)
    x = F.conv3d(hidden_states, p["patch_embedding.weight"], p["patch_embedding.bias"], stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)
    if taps is not None:
        taps["patch"] = x
    temb, tproj, text, image = condition_embed(p, cfg, timestep, encoder_hidden_states, encoder_hidden_states_image)
    tproj = tproj.unflatten(1, (6, -1))
    if image is not None:
        enc = torch.cat([image, text], dim=1)
    else:
        enc = text
    if taps is not None:
        taps["temb"], taps["tproj"], taps["enc"] = temb, tproj, enc
    for i in range(cfg.num_layers):
        x = block_forward(p, i, cfg, x, enc, tproj, rotary, taps, fp8=fp8, fp8_attn=fp8_attn, fp8_linears=fp8_linears)
        if taps is not None:
            taps[f"blocks.{i}.out"] = x
    shift, scale = (p["scale_shift_table"] + temb.unsqueeze(1)).chunk(2, dim=1)
    x = (fp32_layer_norm(x.float(), None, None, cfg.eps) * (1 + scale) + shift).type_as(x)
    x = linear(x, p, "proj_out")
    x = x.reshape(B, ppf, pph, ppw, pt, ph, pw, -1)
    x = x.permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.flatten(6, 7).flatten(4, 5).flatten(2, 3)


# --------------------------------------------------------------------------------------
# synthetic inputs and FLOP accounting
# --------------------------------------------------------------------------------------


def make_synthetic_inputs(cfg: DiTConfig, T: int, h: int, w: int, dtype=torch.float32, text_len: int = 512, real_text: int = 64):
    """Seeds: latents 42 (numpy RandomState, mirrors arch_invariant_rand,
    _ext/imaginaire/utils/misc.py:157-179), text 7, image 11."""
    import numpy as np

    lat = torch.from_numpy(np.random.RandomState(42).standard_normal((1, cfg.in_channels, T, h, w)).astype("float32"))
    text = torch.randn((1, text_len, cfg.text_dim), generator=torch.Generator().manual_seed(7))
    text[:, real_text:] = 0  # zero padding after seq_len, pipeline_chronoedit.py:234-237
    image = None
    if cfg.image_dim is not None:
        image = torch.randn((1, 257, cfg.image_dim), generator=torch.Generator().manual_seed(11)).to(dtype)
    return lat.to(dtype), text.to(dtype), image


def flops_per_forward(cfg: DiTConfig, N: int, Tt: int = 512, Ti: int = 257) -> float:
    """SURVEY.md §8d algorithmic FLOPs per forward."""
    D, Fd, L = cfg.inner_dim, cfg.ffn_dim, cfg.num_layers
    pt, ph, pw = cfg.patch_size
    kin = cfg.in_channels * pt * ph * pw
    per_layer = (
        8 * N * D * D + 4 * N * N * D + 4 * N * D * D + 4 * (Tt + Ti) * D * D + 4 * N * (Tt + Ti) * D + 4 * N * D * Fd
    )
    return L * per_layer + 2 * N * kin * D + 2 * N * D * cfg.out_channels * pt * ph * pw
