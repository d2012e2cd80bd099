"""CPU restatement of the UMT5 encoder the reference uses as its text encoder.  TEST INFRASTRUCTURE ONLY: imported by
tests/ and by the golden generator, never by chronoedit_amd/.

Reference call site: chronoedit_diffusers/pipeline_chronoedit.py:205-243 (``_get_t5_prompt_embeds``):
``self.text_encoder(text_input_ids, mask).last_hidden_state``, rows past each prompt's length zeroed, with ``text_encoder``
a ``transformers.UMT5EncoderModel``.  The arithmetic is in the un-vendored dependency transformers==4.57.1
(requirements_minimal.txt), models/umt5/modeling_umt5.py; restated here from its published structure:
  UMT5LayerNorm.forward          x * rsqrt(mean(x^2) + eps) with fp32 statistics, cast to the weight dtype, times weight
  UMT5Attention.forward          q, k, v, o Linear without bias; scores = q k^T (NO 1/sqrt(d) scaling) + position_bias + mask;
                                 softmax(scores.float()).type_as(scores); every layer owns its relative_attention_bias table
  UMT5Attention._relative_position_bucket / compute_bias   bidirectional log-spaced buckets of (key - query)
  UMT5DenseGatedActDense.forward wo(gelu_new(wi_0(x)) * wi_1(x))
  UMT5Stack.forward              embed_tokens -> blocks -> final_layer_norm; mask = (1 - attention_mask) * finfo(dtype).min
Pinned against the real transformers implementation run in this container: oracle/gen_golden_umt5.py ->
tests/golden/umt5_tiny.pt (tests/test_encoders_oracle.py)."""
import math
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F


@dataclass
class UMT5Cfg:
    vocab_size: int = 256384
    d_model: int = 4096
    d_kv: int = 64
    d_ff: int = 10240
    num_layers: int = 24
    num_heads: int = 64
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6


def param_shapes(cfg: UMT5Cfg) -> Dict[str, tuple]:
    inner = cfg.num_heads * cfg.d_kv
    s = {"shared.weight": (cfg.vocab_size, cfg.d_model), "encoder.final_layer_norm.weight": (cfg.d_model,)}
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer."
        for n in ("q", "k", "v"):
            s[p + f"0.SelfAttention.{n}.weight"] = (inner, cfg.d_model)
        s[p + "0.SelfAttention.o.weight"] = (cfg.d_model, inner)
        s[p + "0.SelfAttention.relative_attention_bias.weight"] = (cfg.relative_attention_num_buckets, cfg.num_heads)
        s[p + "0.layer_norm.weight"] = (cfg.d_model,)
        s[p + "1.DenseReluDense.wi_0.weight"] = (cfg.d_ff, cfg.d_model)
        s[p + "1.DenseReluDense.wi_1.weight"] = (cfg.d_ff, cfg.d_model)
        s[p + "1.DenseReluDense.wo.weight"] = (cfg.d_model, cfg.d_ff)
        s[p + "1.layer_norm.weight"] = (cfg.d_model,)
    return s


def make_synthetic_params(cfg: UMT5Cfg, seed: int = 1357, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif "relative_attention_bias" in k:
            t = 0.5 * torch.randn(shp, generator=g)
        elif k == "shared.weight":
            t = torch.randn(shp, generator=g)
        elif ".q.weight" in k or ".k.weight" in k:
            t = torch.randn(shp, generator=g) * (1.5 / shp[1] ** 0.5) / cfg.d_kv ** 0.25  # keeps un-scaled q.k scores O(1)
        else:
            t = torch.randn(shp, generator=g) / shp[1] ** 0.5
        out[k] = t.to(dtype)
    return out


def relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Encoder (bidirectional) branch of UMT5Attention._relative_position_bucket; rel = key position - query position."""
    nb = num_buckets // 2
    buckets = (rel > 0).to(torch.long) * nb
    rel = torch.abs(rel)
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rel, large)


def t5_layer_norm(x, w, eps):
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    h = x * torch.rsqrt(var + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        h = h.to(w.dtype)
    return w * h


def umt5_encode(params: Dict[str, torch.Tensor], cfg: UMT5Cfg, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """UMT5EncoderModel(input_ids, attention_mask).last_hidden_state; arithmetic in the dtype of params."""
    P = params
    dt = P["shared.weight"].dtype
    B, L = input_ids.shape
    H, dk = cfg.num_heads, cfg.d_kv
    x = P["shared.weight"][input_ids]
    mask_add = (1.0 - attention_mask[:, None, None, :].to(dt)) * torch.finfo(dt).min
    pos = torch.arange(L)
    bucket = relative_position_bucket(pos[None, :] - pos[:, None], cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer."
        h = t5_layer_norm(x, P[p + "0.layer_norm.weight"], cfg.layer_norm_epsilon)
        q = F.linear(h, P[p + "0.SelfAttention.q.weight"]).view(B, L, H, dk).transpose(1, 2)
        k = F.linear(h, P[p + "0.SelfAttention.k.weight"]).view(B, L, H, dk).transpose(1, 2)
        v = F.linear(h, P[p + "0.SelfAttention.v.weight"]).view(B, L, H, dk).transpose(1, 2)
        scores = torch.matmul(q, k.transpose(3, 2))
        bias = P[p + "0.SelfAttention.relative_attention_bias.weight"][bucket].permute(2, 0, 1)[None]
        scores = scores + (bias + mask_add)
        w = F.softmax(scores.float(), dim=-1).type_as(scores)
        a = torch.matmul(w, v).transpose(1, 2).reshape(B, L, H * dk)
        x = x + F.linear(a, P[p + "0.SelfAttention.o.weight"])
        h = t5_layer_norm(x, P[p + "1.layer_norm.weight"], cfg.layer_norm_epsilon)
        g = F.gelu(F.linear(h, P[p + "1.DenseReluDense.wi_0.weight"]), approximate="tanh") * F.linear(h, P[p + "1.DenseReluDense.wi_1.weight"])
        x = x + F.linear(g, P[p + "1.DenseReluDense.wo.weight"])
    return t5_layer_norm(x, P["encoder.final_layer_norm.weight"], cfg.layer_norm_epsilon)


def prompt_embeds(last_hidden_state: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """pipeline_chronoedit.py:231-237: rows past each prompt's length replaced by zeros (same padded length)."""
    out = last_hidden_state.clone()
    lens = attention_mask.gt(0).sum(dim=1)
    for b in range(out.shape[0]):
        out[b, int(lens[b]):] = 0
    return out


def make_synthetic_tokens(cfg: UMT5Cfg, lens, L: int, seed: int = 7):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros((len(lens), L), dtype=torch.long)
    mask = torch.zeros((len(lens), L), dtype=torch.long)
    for b, n in enumerate(lens):
        ids[b, :n] = torch.randint(2, cfg.vocab_size, (n,), generator=g)
        ids[b, n - 1] = 1  # </s>
        mask[b, :n] = 1
    return ids, mask
