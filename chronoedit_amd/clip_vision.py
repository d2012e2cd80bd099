"""Drop-in ``CLIPVisionModel`` (the reference's image encoder) on the gfx950 HIP kernels.

Reference: ``self.image_encoder(**image, output_hidden_states=True).hidden_states[-2]``
(chronoedit_diffusers/pipeline_chronoedit.py:247-256), ``image_encoder = CLIPVisionModel.from_pretrained(...)``
(scripts/run_inference_diffusers.py:333-338).  Same parameter tree as the transformers==4.57.1 class (``vision_model.*`` keys),
same call (``pixel_values=``, ``output_hidden_states=``), same outputs (``last_hidden_state``, ``pooler_output``,
``hidden_states``).  The modules only hold parameters; ``_Engine`` sequences libchronoedit_hip.so launches:
  * patch embedding = ``ce_im2col_patch2d_bf16`` + ``ce_gemm_bf16`` whose residual epilogue adds the position embedding
    (the CLS row ``class_embedding + position[0]`` is a constant, formed when the weights are packed);
  * LayerNorms = ``ce_ln_affine_bf16``; q/k/v as ONE GEMM; MLP GEMMs with the GELU / residual epilogues;
  * attention = ``ce_attention_batched_bf16`` (head_dim 128): ViT-H heads are 80 wide, so every head is zero-padded to
    128 in the packed q/k/v/out weights - the padded q.k terms and the padded v columns are exact zeros - with
    ``softmax_scale = 80^-0.5`` passed explicitly.
No CPU / eager fallback: CPU inputs raise.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import ops, weights

HD_PAD = 128


class _Attn(nn.Module):
    def __init__(self, D, **kw):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(D, D, **kw) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, D, I, **kw):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(D, I, **kw), nn.Linear(I, D, **kw)


class _Layer(nn.Module):
    def __init__(self, D, I, eps, **kw):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(D, eps=eps, **kw)
        self.self_attn = _Attn(D, **kw)
        self.layer_norm2 = nn.LayerNorm(D, eps=eps, **kw)
        self.mlp = _MLP(D, I, **kw)


class _Embeddings(nn.Module):
    def __init__(self, D, C, P, n_pos, **kw):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.zeros(D, **kw))
        self.patch_embedding = nn.Conv2d(C, D, kernel_size=P, stride=P, bias=False, **kw)
        self.position_embedding = nn.Embedding(n_pos, D, **kw)


class _Encoder(nn.Module):
    def __init__(self, L, D, I, eps, **kw):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(D, I, eps, **kw) for _ in range(L)])


class _VisionTransformer(nn.Module):
    def __init__(self, cfg, **kw):
        super().__init__()
        n_pos = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.embeddings = _Embeddings(cfg.hidden_size, cfg.num_channels, cfg.patch_size, n_pos, **kw)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps, **kw)  # (sic) the transformers spelling
        self.encoder = _Encoder(cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps, **kw)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps, **kw)


_ACT_EPI = {"gelu": ops.EPI_BIAS_GELU_ERF, "gelu_pytorch_tanh": ops.EPI_BIAS_GELU, "gelu_new": ops.EPI_BIAS_GELU}


class CLIPVisionModel(nn.Module):
    def __init__(self, hidden_size: int = 1280, intermediate_size: int = 5120, num_hidden_layers: int = 32, num_attention_heads: int = 16,
                 image_size: int = 224, patch_size: int = 14, num_channels: int = 3, layer_norm_eps: float = 1e-5, hidden_act: str = "gelu",
                 device=None, dtype: torch.dtype = torch.bfloat16, **unused):
        super().__init__()
        if hidden_act not in _ACT_EPI:
            raise NotImplementedError(f"hidden_act={hidden_act!r}: the GEMM epilogues provide {sorted(_ACT_EPI)}")
        if hidden_size % num_attention_heads or hidden_size // num_attention_heads > HD_PAD:
            raise NotImplementedError("head_dim must divide hidden_size and be <= 128 (heads are zero-padded to the kernel's 128)")
        if hidden_size % 64 or intermediate_size % 64:
            raise NotImplementedError("hidden_size and intermediate_size must be multiples of 64 (GEMM K tile)")
        self.config = SimpleNamespace(hidden_size=hidden_size, intermediate_size=intermediate_size, num_hidden_layers=num_hidden_layers,
                                      num_attention_heads=num_attention_heads, image_size=image_size, patch_size=patch_size,
                                      num_channels=num_channels, layer_norm_eps=layer_norm_eps, hidden_act=hidden_act)
        self.vision_model = _VisionTransformer(self.config, device=device, dtype=dtype)
        self._engine: Optional[_Engine] = None

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None, torch_dtype: torch.dtype = torch.bfloat16, device=None, **unused):
        """`CLIPVisionModel.from_pretrained(model_path, subfolder="image_encoder", torch_dtype=torch.float32)` is how the reference
        runner loads it (run_inference_diffusers.py:333-337).  The engine computes in bf16: an fp32 request is honoured as "read the
        checkpoint, round every parameter to bf16 once" (the pipeline casts the embeddings to the transformer's bf16 anyway,
        pipeline_chronoedit.py:661); the request is remembered in `requested_dtype`."""
        cfg = weights.read_config(path, subfolder)
        cfg = cfg.get("vision_config", cfg)  # a full CLIP config nests the tower's
        model = cls(**cfg, device=device, dtype=torch.bfloat16)
        model.requested_dtype = torch_dtype
        files = weights.shard_files(path, subfolder, names=("model.safetensors",))
        sd = weights.load_state_dict_files(files)
        sd = {k: v for k, v in sd.items() if k.startswith("vision_model.")}  # a full CLIP checkpoint also carries the text tower
        weights.assign_state_dict(model, sd, ignore_unexpected=(r"position_ids$",))
        return model

    @property
    def dtype(self):
        return self.vision_model.pre_layrnorm.weight.dtype

    @property
    def device(self):
        return self.vision_model.pre_layrnorm.weight.device

    def invalidate(self):
        self._engine = None

    def _apply(self, fn, *a, **kw):
        self._engine = None
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._engine = None
        return super().load_state_dict(*a, **kw)

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, output_hidden_states: bool = False, return_dict: bool = True, **unused):
        if not pixel_values.is_cuda:
            raise ops.HipKernelError("CLIPVisionModel (chronoedit_amd) runs only on an MI355X device: there is no CPU fallback")
        if self._engine is None:
            self._engine = _Engine(self)
        hs = self._engine.forward(pixel_values)
        B = pixel_values.shape[0]
        T, D = self._engine.T, self.config.hidden_size
        last = hs[-1].view(B, T, D)
        pooled = self._engine.pool(hs[-1]).view(B, D)
        out = SimpleNamespace(last_hidden_state=last, pooler_output=pooled,
                              hidden_states=tuple(h.view(B, T, D) for h in hs) if output_hidden_states else None)
        return out if return_dict else (last, pooled) + ((out.hidden_states,) if output_hidden_states else ())


class _Engine:
    def __init__(self, model: CLIPVisionModel):
        c = model.config
        vm = model.vision_model
        self.cfg, self.dev = c, model.device
        if model.dtype != torch.bfloat16:
            raise ops.HipKernelError("the HIP path computes in bf16: load the encoder with torch_dtype=torch.bfloat16")
        D, H = c.hidden_size, c.num_attention_heads
        hd = D // H
        self.D, self.H, self.hd = D, H, hd
        self.T = (c.image_size // c.patch_size) ** 2 + 1
        f32 = lambda t: t.detach().float().contiguous()
        kraw = c.num_channels * c.patch_size ** 2
        self.kpad = (kraw + 63) // 64 * 64
        w = torch.zeros((D, self.kpad), dtype=torch.bfloat16, device=self.dev)
        w[:, :kraw] = vm.embeddings.patch_embedding.weight.detach().reshape(D, kraw)
        self.w_patch = w
        pos = vm.embeddings.position_embedding.weight.detach()
        self.cls_row = (vm.embeddings.class_embedding.detach() + pos[0]).contiguous()  # bf16 add, as cat(...) + position does
        self.pos_rest = pos[1:].contiguous()
        self.pre = (f32(vm.pre_layrnorm.weight), f32(vm.pre_layrnorm.bias))
        self.post = (f32(vm.post_layernorm.weight), f32(vm.post_layernorm.bias))
        self.act_epi = _ACT_EPI[c.hidden_act]

        def pad_rows(wt, b):  # [H*hd, D] -> [H*128, D]: head h occupies rows [128 h, 128 h + hd), the rest zero
            wp = torch.zeros((H, HD_PAD, wt.shape[1]), dtype=wt.dtype, device=self.dev)
            wp[:, :hd] = wt.detach().view(H, hd, -1)
            bp = torch.zeros((H, HD_PAD), dtype=torch.float32, device=self.dev)
            bp[:, :hd] = b.detach().float().view(H, hd)
            return wp.view(H * HD_PAD, -1), bp.view(-1)

        self.layers = []
        for lyr in vm.encoder.layers:
            p = SimpleNamespace()
            p.ln1 = (f32(lyr.layer_norm1.weight), f32(lyr.layer_norm1.bias))
            p.ln2 = (f32(lyr.layer_norm2.weight), f32(lyr.layer_norm2.bias))
            a = lyr.self_attn
            qw, qb = pad_rows(a.q_proj.weight, a.q_proj.bias)
            kw, kb = pad_rows(a.k_proj.weight, a.k_proj.bias)
            vw, vb = pad_rows(a.v_proj.weight, a.v_proj.bias)
            p.w_qkv, p.b_qkv = torch.cat([qw, kw, vw]).contiguous(), torch.cat([qb, kb, vb]).contiguous()
            ow = torch.zeros((D, H, HD_PAD), dtype=torch.bfloat16, device=self.dev)
            ow[:, :, :hd] = a.out_proj.weight.detach().view(D, H, hd)
            p.w_o, p.b_o = ow.view(D, H * HD_PAD).contiguous(), f32(a.out_proj.bias)
            p.w1, p.b1 = lyr.mlp.fc1.weight.detach().contiguous(), f32(lyr.mlp.fc1.bias)
            p.w2, p.b2 = lyr.mlp.fc2.weight.detach().contiguous(), f32(lyr.mlp.fc2.bias)
            self.layers.append(p)

    def forward(self, pixel_values: torch.Tensor):
        c, D, H, T = self.cfg, self.D, self.H, self.T
        B, C, Hh, Ww = pixel_values.shape
        if (Hh, Ww) != (c.image_size, c.image_size) or C != c.num_channels:
            raise ValueError(f"Input image size ({Hh}*{Ww}) doesn't match model ({c.image_size}*{c.image_size}).")  # CLIPVisionEmbeddings
        e = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=self.dev)
        cols = ops.im2col_patch2d(pixel_values.to(torch.bfloat16).contiguous(), c.patch_size, self.kpad)
        n_p = T - 1
        emb = e(B * T, D)
        for b in range(B):
            emb[b * T].copy_(self.cls_row)
            ops.gemm(cols[b * n_p:(b + 1) * n_p], self.w_patch, None, out=emb[b * T + 1:(b + 1) * T], epilogue=ops.EPI_GATE_RES,
                     gate=None, res=self.pos_rest)
        x = ops.ln_affine(emb, self.pre[0], self.pre[1], c.layer_norm_eps)
        hs = [x]
        y, qkv, att = e(B * T, D), e(B * T, 3 * H * HD_PAD), e(B * T, H * HD_PAD)
        f = e(B * T, c.intermediate_size)
        Dp = H * HD_PAD
        for p in self.layers:
            ops.ln_affine(x, p.ln1[0], p.ln1[1], c.layer_norm_eps, out=y)
            ops.gemm(y, p.w_qkv, p.b_qkv, out=qkv)
            ops.attention(qkv[:, :Dp], qkv[:, Dp:2 * Dp], qkv[:, 2 * Dp:], H, out=att, scale=self.hd ** -0.5, batch=B)
            x1 = ops.gemm(att, p.w_o, p.b_o, epilogue=ops.EPI_GATE_RES, gate=None, res=x)
            ops.ln_affine(x1, p.ln2[0], p.ln2[1], c.layer_norm_eps, out=y)
            ops.gemm(y, p.w1, p.b1, out=f, epilogue=self.act_epi)
            x = ops.gemm(f, p.w2, p.b2, epilogue=ops.EPI_GATE_RES, gate=None, res=x1)
            hs.append(x)
        return hs

    def pool(self, last: torch.Tensor) -> torch.Tensor:
        cls_rows = last.view(-1, self.T, self.D)[:, 0].contiguous()
        return ops.ln_affine(cls_rows, self.post[0], self.post[1], self.cfg.layer_norm_eps)
