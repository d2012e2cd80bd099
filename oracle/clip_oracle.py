"""CPU restatement of the CLIP vision tower the reference uses as its image encoder.  TEST INFRASTRUCTURE ONLY: imported by
tests/ and by the golden generator, never by chronoedit_amd/.

Reference call site: chronoedit_diffusers/pipeline_chronoedit.py:247-256 (``encode_image``):
``self.image_encoder(**image, output_hidden_states=True).hidden_states[-2]`` with ``image_encoder`` a
``transformers.CLIPVisionModel`` (run_inference_diffusers.py:333-338).  The arithmetic is in the un-vendored dependency
transformers==4.57.1 (requirements_minimal.txt), models/clip/modeling_clip.py; restated here from its published structure:
  CLIPVisionEmbeddings.forward   patch Conv2d(k = s = patch, no bias) -> [B, n, D]; cat(class_embedding, patches) + position_embedding
  CLIPVisionTransformer.forward  pre_layrnorm -> encoder layers -> (post_layernorm only on the pooled CLS token)
  CLIPEncoderLayer.forward       x + attn(LN1(x)); x + fc2(act(fc1(LN2(x))))
  CLIPAttention.forward          q/k/v/out Linear with bias, softmax(q k^T * head_dim^-0.5) v per head
``hidden_states`` = (embeddings after pre_layrnorm, layer 1 output, ..., layer L output); the pipeline takes [-2].
Pinned against the real transformers implementation run in this container: oracle/gen_golden_clip.py ->
tests/golden/clip_tiny.pt (tests/test_encoders_oracle.py)."""
from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class CLIPVisionCfg:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    hidden_act: str = "gelu"


def param_shapes(cfg: CLIPVisionCfg) -> Dict[str, tuple]:
    D, I = cfg.hidden_size, cfg.intermediate_size
    n_pos = (cfg.image_size // cfg.patch_size) ** 2 + 1
    s = {
        "vision_model.embeddings.class_embedding": (D,),
        "vision_model.embeddings.patch_embedding.weight": (D, cfg.num_channels, cfg.patch_size, cfg.patch_size),
        "vision_model.embeddings.position_embedding.weight": (n_pos, D),
        "vision_model.pre_layrnorm.weight": (D,), "vision_model.pre_layrnorm.bias": (D,),
        "vision_model.post_layernorm.weight": (D,), "vision_model.post_layernorm.bias": (D,),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"vision_model.encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            s[p + ln + ".weight"] = (D,)
            s[p + ln + ".bias"] = (D,)
        for lin in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{lin}.weight"] = (D, D)
            s[p + f"self_attn.{lin}.bias"] = (D,)
        s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"] = (I, D), (I,)
        s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"] = (D, I), (D,)
    return s


def make_synthetic_params(cfg: CLIPVisionCfg, seed: int = 2468, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if "norm" in k and k.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif "embedding" in k and "patch" not in k:
            t = 0.3 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g) / fan_in ** 0.5
        out[k] = t.to(dtype)
    return out


def _act(x, name):
    if name == "gelu":
        return F.gelu(x)
    if name in ("gelu_pytorch_tanh", "gelu_new"):
        return F.gelu(x, approximate="tanh")
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(name)


def clip_vision_hidden_states(params: Dict[str, torch.Tensor], cfg: CLIPVisionCfg, pixel_values: torch.Tensor) -> List[torch.Tensor]:
    """hidden_states tuple of CLIPVisionModel(pixel_values, output_hidden_states=True); arithmetic in the dtype of params."""
    P = params
    dt = P["vision_model.embeddings.patch_embedding.weight"].dtype
    B = pixel_values.shape[0]
    D, H = cfg.hidden_size, cfg.num_attention_heads
    hd = D // H
    x = F.conv2d(pixel_values.to(dt), P["vision_model.embeddings.patch_embedding.weight"], stride=cfg.patch_size)  # modeling_clip: Embeddings
    x = x.flatten(2).transpose(1, 2)
    cls = P["vision_model.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + P["vision_model.embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (D,), P["vision_model.pre_layrnorm.weight"], P["vision_model.pre_layrnorm.bias"], cfg.layer_norm_eps)
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        p = f"vision_model.encoder.layers.{i}."
        y = F.layer_norm(x, (D,), P[p + "layer_norm1.weight"], P[p + "layer_norm1.bias"], cfg.layer_norm_eps)
        q = F.linear(y, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"]).view(B, -1, H, hd).transpose(1, 2)
        k = F.linear(y, P[p + "self_attn.k_proj.weight"], P[p + "self_attn.k_proj.bias"]).view(B, -1, H, hd).transpose(1, 2)
        v = F.linear(y, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"]).view(B, -1, H, hd).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v, scale=hd ** -0.5)  # CLIPAttention: scaling = head_dim ** -0.5
        a = a.transpose(1, 2).reshape(B, -1, D)
        x = x + F.linear(a, P[p + "self_attn.out_proj.weight"], P[p + "self_attn.out_proj.bias"])
        y = F.layer_norm(x, (D,), P[p + "layer_norm2.weight"], P[p + "layer_norm2.bias"], cfg.layer_norm_eps)
        y = _act(F.linear(y, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"]), cfg.hidden_act)
        x = x + F.linear(y, P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"])
        hs.append(x)
    return hs


def make_synthetic_pixels(cfg: CLIPVisionCfg, batch: int = 1, seed: int = 11) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, cfg.num_channels, cfg.image_size, cfg.image_size, generator=g)
