"""Drop-in ``UMT5EncoderModel`` (the reference's text encoder) on the gfx950 HIP kernels.

Reference: ``self.text_encoder(text_input_ids, mask).last_hidden_state`` and the zeroing of the rows past each prompt's
length (chronoedit_diffusers/pipeline_chronoedit.py:205-243).  Same parameter tree as the transformers==4.57.1 class
(``shared`` / ``encoder.block.N.layer.0.SelfAttention.{q,k,v,o,relative_attention_bias}`` / ``layer.1.DenseReluDense.{wi_0,wi_1,wo}``
/ ``layer_norm`` keys), same call, ``last_hidden_state`` out.  The modules only hold parameters; ``_Engine`` sequences
libchronoedit_hip.so launches per layer:
  ce_rmsnorm_bf16 -> ce_gemm_bf16 (q|k fused, no bias) and ce_gemm_batched_bf16 computing V^T = W_v X^T directly (the P.V
  product wants V with keys contiguous; swapping the GEMM operand roles yields that layout for free) -> per-head
  Q K^T in fp32 (ce_gemm_batched_bf16, NO 1/sqrt(d) scaling in UMT5) -> ce_softmax_t5_bf16 (relative-position bias from the
  layer's bucket table + key padding mask + fp32 softmax) -> per-head P V (ce_gemm_batched_bf16) -> o projection with the
  residual epilogue; ce_rmsnorm_bf16 -> wi_0 with the tanh-GELU epilogue, wi_1 with the multiply epilogue, wo with the
  residual epilogue.
Scores stay fp32 between the product and the softmax (the reference rounds them to bf16 twice on the way).
No CPU / eager fallback: CPU inputs raise.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import ops, weights


class _T5Norm(nn.Module):
    def __init__(self, D, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(D, **kw))


class _SelfAttention(nn.Module):
    def __init__(self, D, inner, buckets, heads, **kw):
        super().__init__()
        self.q, self.k, self.v = (nn.Linear(D, inner, bias=False, **kw) for _ in range(3))
        self.o = nn.Linear(inner, D, bias=False, **kw)
        self.relative_attention_bias = nn.Embedding(buckets, heads, **kw)


class _LayerSelfAttention(nn.Module):
    def __init__(self, D, inner, buckets, heads, **kw):
        super().__init__()
        self.SelfAttention = _SelfAttention(D, inner, buckets, heads, **kw)
        self.layer_norm = _T5Norm(D, **kw)


class _Dense(nn.Module):
    def __init__(self, D, F, **kw):
        super().__init__()
        self.wi_0, self.wi_1 = nn.Linear(D, F, bias=False, **kw), nn.Linear(D, F, bias=False, **kw)
        self.wo = nn.Linear(F, D, bias=False, **kw)


class _LayerFF(nn.Module):
    def __init__(self, D, F, **kw):
        super().__init__()
        self.DenseReluDense = _Dense(D, F, **kw)
        self.layer_norm = _T5Norm(D, **kw)


class _Block(nn.Module):
    def __init__(self, D, inner, F, buckets, heads, **kw):
        super().__init__()
        self.layer = nn.ModuleList([_LayerSelfAttention(D, inner, buckets, heads, **kw), _LayerFF(D, F, **kw)])


class _Stack(nn.Module):
    def __init__(self, cfg, embed, **kw):
        super().__init__()
        self.embed_tokens = embed
        inner = cfg.num_heads * cfg.d_kv
        self.block = nn.ModuleList([_Block(cfg.d_model, inner, cfg.d_ff, cfg.relative_attention_num_buckets, cfg.num_heads, **kw)
                                    for _ in range(cfg.num_layers)])
        self.final_layer_norm = _T5Norm(cfg.d_model, **kw)


def relative_position_buckets(Lq: int, Lk: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """int32 LUT indexed by (key - query + Lq - 1): the bidirectional bucket of UMT5Attention._relative_position_bucket
    (modeling_umt5.py, encoder branch: half the buckets per sign, half of those exact, the rest log-spaced up to max_distance)."""
    rel = torch.arange(-(Lq - 1), Lk, dtype=torch.long)
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    a = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.clamp(large, max=nb - 1)
    return (out + torch.where(a < max_exact, a, large)).to(torch.int32)


class UMT5EncoderModel(nn.Module):
    def __init__(self, vocab_size: int = 256384, d_model: int = 4096, d_kv: int = 64, d_ff: int = 10240, num_layers: int = 24,
                 num_heads: int = 64, relative_attention_num_buckets: int = 32, relative_attention_max_distance: int = 128,
                 layer_norm_epsilon: float = 1e-6, feed_forward_proj: str = "gated-gelu", device=None,
                 dtype: torch.dtype = torch.bfloat16, **unused):
        super().__init__()
        if feed_forward_proj != "gated-gelu":
            raise NotImplementedError("only the gated-gelu feed-forward of UMT5 is implemented")
        if d_kv != 64 or d_model % 64 or d_ff % 64:
            raise NotImplementedError("d_kv must be 64 and d_model, d_ff multiples of 64 (GEMM K tile)")
        self.config = SimpleNamespace(vocab_size=vocab_size, d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=num_layers,
                                      num_heads=num_heads, relative_attention_num_buckets=relative_attention_num_buckets,
                                      relative_attention_max_distance=relative_attention_max_distance,
                                      layer_norm_epsilon=layer_norm_epsilon, feed_forward_proj=feed_forward_proj)
        kw = dict(device=device, dtype=dtype)
        self.shared = nn.Embedding(vocab_size, d_model, **kw)
        self.encoder = _Stack(self.config, self.shared, **kw)  # embed_tokens is tied to shared, as in transformers
        self._engine: Optional[_Engine] = None

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype: torch.dtype = torch.bfloat16, device=None, **unused):
        cfg = weights.read_config(path, subfolder)
        model = cls(**cfg, device=device, dtype=torch_dtype)
        sd = weights.load_state_dict_files(weights.shard_files(path, subfolder, names=("model.safetensors",)))
        sd.pop("encoder.embed_tokens.weight", None)  # tied to shared.weight
        weights.assign_state_dict(model, sd, ignore_unexpected=(r"^decoder\.", r"^lm_head\."))
        return model

    @property
    def dtype(self):
        return self.shared.weight.dtype

    @property
    def device(self):
        return self.shared.weight.device

    def invalidate(self):
        self._engine = None

    def _apply(self, fn, *a, **kw):
        self._engine = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._engine = None
        return super().load_state_dict(*a, **kw)

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, return_dict: bool = True, **unused):
        if not input_ids.is_cuda:
            raise ops.HipKernelError("UMT5EncoderModel (chronoedit_amd) runs only on an MI355X device: there is no CPU fallback")
        if self._engine is None:
            self._engine = _Engine(self)
        out = self._engine.forward(input_ids, attention_mask)
        return SimpleNamespace(last_hidden_state=out) if return_dict else (out,)


class _Engine:
    def __init__(self, model: UMT5EncoderModel):
        c = model.config
        self.cfg, self.dev = c, model.device
        if model.dtype != torch.bfloat16:
            raise ops.HipKernelError("the HIP path computes in bf16: load the encoder with torch_dtype=torch.bfloat16")
        self.emb = model.shared.weight.detach()
        self.final_w = model.encoder.final_layer_norm.weight.detach().contiguous()
        self.layers = []
        for blk in model.encoder.block:
            a, ff = blk.layer[0], blk.layer[1]
            p = SimpleNamespace()
            sa = a.SelfAttention
            p.ln0 = a.layer_norm.weight.detach().contiguous()
            p.w_qk = torch.cat([sa.q.weight.detach(), sa.k.weight.detach()]).contiguous()
            p.w_v = sa.v.weight.detach().contiguous()
            p.w_o = sa.o.weight.detach().contiguous()
            p.table = sa.relative_attention_bias.weight.detach().float().contiguous()  # [buckets, heads]
            p.ln1 = ff.layer_norm.weight.detach().contiguous()
            d = ff.DenseReluDense
            p.w_i0, p.w_i1, p.w_o2 = d.wi_0.weight.detach().contiguous(), d.wi_1.weight.detach().contiguous(), d.wo.weight.detach().contiguous()
            self.layers.append(p)
        self._lut = {}

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]):
        c = self.cfg
        if input_ids.dim() != 2 or input_ids.dtype != torch.int64:
            raise ValueError("input_ids must be an int64 [batch, length] tensor")
        B, L = input_ids.shape
        if L % 8 or L > 1024:
            raise ValueError(f"sequence length {L}: must be a multiple of 8 and <= 1024 (the pipeline pads prompts to 512)")
        if attention_mask is None:
            valid = torch.full((B,), L, dtype=torch.int32, device=self.dev)
        else:
            m = attention_mask.to(self.dev).gt(0)
            lens = m.sum(dim=1)
            if not torch.equal(m, torch.arange(L, device=self.dev)[None, :] < lens[:, None]) or int(lens.min()) < 1:
                raise ValueError("attention_mask must mark a non-empty prefix of every row (right padding, as the tokenizer produces)")
            valid = lens.to(torch.int32).contiguous()
        D, H, dk, F = c.d_model, c.num_heads, c.d_kv, c.d_ff
        inner = H * dk
        Lp = (L + 63) // 64 * 64
        if L not in self._lut:
            self._lut = {L: relative_position_buckets(L, L, c.relative_attention_num_buckets, c.relative_attention_max_distance).to(self.dev)}
        lut = self._lut[L]
        e = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=self.dev)
        x = ops.gather_rows(self.emb, input_ids.reshape(-1).contiguous())
        h, qk, att = e(B * L, D), e(B * L, 2 * inner), e(B * L, inner)
        vt = torch.zeros((B * inner, Lp), dtype=torch.bfloat16, device=self.dev)  # columns >= L stay zero (K padding of the P.V product)
        scores = torch.empty((B * H * L, L), dtype=torch.float32, device=self.dev)
        probs = e(B * H * L, Lp)
        g0, g = e(B * L, F), e(B * L, F)
        for p in self.layers:
            ops.rmsnorm(x, p.ln0, c.layer_norm_epsilon, out=h)
            ops.gemm(h, p.w_qk, None, out=qk)
            ops.gemm_batched(p.w_v, h, vt, M=inner, N=L, K=D, lda=D, ldw=D, ldc=Lp, batch=(B, 1),
                             stride_a=(0, 0), stride_w=(L * D, 0), stride_c=(inner * Lp, 0))
            ops.gemm_batched(qk, qk[:, inner:], scores, M=L, N=L, K=dk, lda=2 * inner, ldw=2 * inner, ldc=L, batch=(H, B),
                             stride_a=(dk, L * 2 * inner), stride_w=(dk, L * 2 * inner), stride_c=(L * L, H * L * L), f32_out=True)
            ops.softmax_t5(scores, probs, B, H, L, L, lut, p.table, valid)
            ops.gemm_batched(probs, vt, att, M=L, N=dk, K=Lp, lda=Lp, ldw=Lp, ldc=inner, batch=(H, B),
                             stride_a=(L * Lp, H * L * Lp), stride_w=(dk * Lp, inner * Lp), stride_c=(dk, L * inner))
            ops.gemm(att, p.w_o, None, out=x, epilogue=ops.EPI_GATE_RES, gate=None, res=x)
            ops.rmsnorm(x, p.ln1, c.layer_norm_epsilon, out=h)
            ops.gemm(h, p.w_i0, None, out=g0, epilogue=ops.EPI_BIAS_GELU)
            ops.gemm(h, p.w_i1, None, out=g, epilogue=ops.EPI_MUL, res=g0)
            ops.gemm(g, p.w_o2, None, out=x, epilogue=ops.EPI_GATE_RES, gate=None, res=x)
        return ops.rmsnorm(x, self.final_w, c.layer_norm_epsilon).view(B, L, D)


def t5_prompt_embeds(text_encoder: UMT5EncoderModel, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """``_get_t5_prompt_embeds`` after the tokenizer (pipeline_chronoedit.py:231-237): encoder output with the rows past each
    prompt's length set to zero, same padded length."""
    out = text_encoder(input_ids, attention_mask).last_hidden_state
    keep = attention_mask.to(out.device).gt(0)[:, :, None]
    return out * keep.to(out.dtype)
