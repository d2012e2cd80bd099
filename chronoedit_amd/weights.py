"""Checkpoint plumbing of the drop-in transformer: diffusers-layout safetensors shards in, LoRA adapters fused into the
weights before the engine packs them for the HIP kernels.

Mirrors what the reference's runner does through diffusers / PEFT (scripts/run_inference_diffusers.py:349-376):
``ChronoEditTransformer3DModel.from_pretrained(path, subfolder="transformer", torch_dtype=bf16)``, then
``pipe.load_lora_weights(file, adapter_name=...)`` and ``pipe.fuse_lora(adapter_names=[...], lora_scale=s)``.  Fusing
means the distilled / upscaler LoRAs cost nothing at run time: the GEMM kernels only ever see ``W + s (alpha/r) B A``.
Host-side, one-off work; nothing here is on the per-step path.
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Iterable, List, Optional, Tuple

import torch

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
INDEX_NAME = WEIGHTS_NAME + ".index.json"
CONFIG_NAME = "config.json"


def _resolve(path: str, subfolder: Optional[str]) -> str:
    d = os.path.join(path, subfolder) if subfolder else path
    if not os.path.isdir(d):
        raise FileNotFoundError(f"no checkpoint directory {d}")
    return d


def read_config(path: str, subfolder: Optional[str] = None) -> Dict:
    """config.json of a diffusers model directory, without the bookkeeping keys (``_class_name`` ...)."""
    with open(os.path.join(_resolve(path, subfolder), CONFIG_NAME)) as f:
        cfg = json.load(f)
    return {k: v for k, v in cfg.items() if not k.startswith("_")}


def shard_files(path: str, subfolder: Optional[str] = None, names: Iterable[str] = (WEIGHTS_NAME,)) -> List[str]:
    """The safetensors files that make up the checkpoint: ``<name>`` or the shards listed by ``<name>.index.json``, for the
    first of ``names`` that exists (diffusers models: diffusion_pytorch_model.safetensors; transformers: model.safetensors)."""
    d = _resolve(path, subfolder)
    for name in names:
        index = os.path.join(d, name + ".index.json")
        if os.path.exists(index):
            with open(index) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
            return [os.path.join(d, n) for n in shards]
        single = os.path.join(d, name)
        if os.path.exists(single):
            return [single]
    raise FileNotFoundError(f"none of {[n for n in names]} (or their .index.json) under {d}")


def load_state_dict_files(files: Iterable[str], device: str = "cpu") -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    sd: Dict[str, torch.Tensor] = {}
    for fn in files:
        with safe_open(fn, framework="pt", device=device) as f:
            for k in f.keys():
                if k in sd:
                    raise ValueError(f"tensor {k} appears in more than one shard")
                sd[k] = f.get_tensor(k)
    return sd


def assign_state_dict(model: torch.nn.Module, sd: Dict[str, torch.Tensor], ignore_unexpected: Iterable[str] = ()) -> None:
    """Copy a checkpoint into the parameter tree.  Every parameter keeps the dtype the constructor gave it - that is how
    the reference's ``_keep_in_fp32_modules`` (transformer_chronoedit.py:338) is honoured under ``torch_dtype=bf16``.
    Unexpected keys raise unless they match ``ignore_unexpected`` (``norm_added_q``: transformer_chronoedit.py:339)."""
    own = dict(model.named_parameters())
    pats = [re.compile(p) for p in ignore_unexpected]
    unexpected = [k for k in sd if k not in own and not any(p.search(k) for p in pats)]
    missing = [k for k in own if k not in sd]
    if missing or unexpected:
        raise KeyError(f"checkpoint does not match the model: missing {sorted(missing)[:5]} unexpected {sorted(unexpected)[:5]}")
    with torch.no_grad():
        for k, p in own.items():
            t = sd[k]
            if tuple(t.shape) != tuple(p.shape):
                raise ValueError(f"{k}: checkpoint shape {tuple(t.shape)} != parameter shape {tuple(p.shape)}")
            p.copy_(t.to(device=p.device, dtype=p.dtype))


def save_pretrained(model: torch.nn.Module, path: str, config: Dict, max_shard_bytes: int = 5 << 30, class_name: str = "") -> List[str]:
    """Write config.json + safetensors shard(s) (+ index) in the diffusers layout.  Used by the tests and by tools that
    need a checkpoint directory; returns the shard file names."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = dict(config)
    cfg["_class_name"] = class_name or type(model).__name__
    with open(os.path.join(path, CONFIG_NAME), "w") as f:
        json.dump(cfg, f, indent=2, default=lambda o: list(o))
    shards: List[Dict[str, torch.Tensor]] = [{}]
    size = 0
    for k, p in model.named_parameters():
        nbytes = p.numel() * p.element_size()
        if shards[-1] and size + nbytes > max_shard_bytes:
            shards.append({})
            size = 0
        shards[-1][k] = p.detach().to("cpu").contiguous()
        size += nbytes
    if len(shards) == 1:
        save_file(shards[0], os.path.join(path, WEIGHTS_NAME))
        return [WEIGHTS_NAME]
    names, weight_map = [], {}
    for i, sh in enumerate(shards):
        name = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(path, name))
        names.append(name)
        weight_map.update({k: name for k in sh})
    with open(os.path.join(path, INDEX_NAME), "w") as f:
        json.dump({"metadata": {"total_size": sum(p.numel() * p.element_size() for p in model.parameters())},
                   "weight_map": weight_map}, f, indent=2)
    return names


# ------------------------------------------------------------------------------------------
# Wan-native checkpoint names (the two sibling stacks of the reference)
# ------------------------------------------------------------------------------------------
# chronoedit_diffsynth.WanModel and chronoedit/_src wan2pt1 keep the original Wan names; the reference converts diffusers ->
# native in WanModelStateDictConverter.from_diffusers (wan_video_dit_chronoedit.py:434-505).  This is the same correspondence
# written from the module structure (attn1 = self_attn, attn2 = cross_attn with k_img / v_img for the image tokens, FeedForward
# = ffn.0 / ffn.2, norm2 = norm3, scale_shift_table = modulation, condition embedder = text_embedding / time_embedding /
# time_projection / img_emb.proj, proj_out + scale_shift_table = head), usable in both directions so that a native
# checkpoint loads into the drop-in transformer and a fused one can be written back for those stacks.
_BLOCK_RULES = [
    (r"attn1\.to_(q|k|v)\.", r"self_attn.\1."), (r"attn1\.to_out\.0\.", "self_attn.o."), (r"attn1\.norm_(q|k)\.", r"self_attn.norm_\1."),
    (r"attn2\.to_(q|k|v)\.", r"cross_attn.\1."), (r"attn2\.to_out\.0\.", "cross_attn.o."), (r"attn2\.norm_(q|k)\.", r"cross_attn.norm_\1."),
    (r"attn2\.add_(k|v)_proj\.", r"cross_attn.\1_img."), (r"attn2\.norm_added_k\.", "cross_attn.norm_k_img."),
    (r"ffn\.net\.0\.proj\.", "ffn.0."), (r"ffn\.net\.2\.", "ffn.2."), (r"norm2\.", "norm3."), (r"scale_shift_table$", "modulation"),
]
_TOP_RULES = [
    (r"^condition_embedder\.text_embedder\.linear_1\.", "text_embedding.0."), (r"^condition_embedder\.text_embedder\.linear_2\.", "text_embedding.2."),
    (r"^condition_embedder\.time_embedder\.linear_1\.", "time_embedding.0."), (r"^condition_embedder\.time_embedder\.linear_2\.", "time_embedding.2."),
    (r"^condition_embedder\.time_proj\.", "time_projection.1."),
    (r"^condition_embedder\.image_embedder\.norm1\.", "img_emb.proj.0."), (r"^condition_embedder\.image_embedder\.ff\.net\.0\.proj\.", "img_emb.proj.1."),
    (r"^condition_embedder\.image_embedder\.ff\.net\.2\.", "img_emb.proj.3."), (r"^condition_embedder\.image_embedder\.norm2\.", "img_emb.proj.4."),
    (r"^scale_shift_table$", "head.modulation"), (r"^proj_out\.", "head.head."), (r"^patch_embedding\.", "patch_embedding."),
]


def diffusers_to_wan_native_key(key: str) -> str:
    m = re.match(r"^blocks\.(\d+)\.(.*)$", key)
    if m:
        rest = m.group(2)
        for pat, rep in _BLOCK_RULES:
            new, n = re.subn("^" + pat, rep, rest)
            if n:
                return f"blocks.{m.group(1)}.{new}"
        raise KeyError(f"no Wan-native name for {key}")
    for pat, rep in _TOP_RULES:
        new, n = re.subn(pat, rep, key)
        if n:
            return new
    raise KeyError(f"no Wan-native name for {key}")


def wan_native_to_diffusers(sd: Dict[str, torch.Tensor], diffusers_keys: Iterable[str]) -> Dict[str, torch.Tensor]:
    """Rename a Wan-native state dict (diffsynth ``WanModel`` / ``wan2pt1`` names) to the diffusers names of the drop-in
    transformer.  ``diffusers_keys`` = the model's parameter names (the map is built from them, so nothing is guessed)."""
    back = {diffusers_to_wan_native_key(k): k for k in diffusers_keys}
    out, unknown = {}, []
    for k, v in sd.items():
        if k in back:
            out[back[k]] = v
        elif k in back.values():  # already a diffusers name
            out[k] = v
        else:
            unknown.append(k)
    if unknown:
        raise KeyError(f"keys that are neither Wan-native nor diffusers names of this model: {sorted(unknown)[:5]}")
    return out


# ------------------------------------------------------------------------------------------
# Wan VAE: diffusers AutoencoderKLWan names -> the native names of _src/tokenizers/wan2pt1.py that chronoedit_amd.vae uses
# ------------------------------------------------------------------------------------------
def wan_vae_diffusers_to_native(sd: Dict[str, torch.Tensor], num_res_blocks: int = 2, num_stages: int = 4) -> Dict[str, torch.Tensor]:
    """Rename a diffusers ``AutoencoderKLWan`` state dict (what ``from_pretrained(subfolder="vae")`` reads,
    run_inference_diffusers.py:341-345) to the native Wan names.  diffusers is NOT installed in this environment, so the
    diffusers side of this table is written from its published layout (conv_in / down_blocks (flat) / mid_block.{resnets,
    attentions} / norm_out / conv_out, decoder up_blocks.{i}.{resnets.{j}, upsamplers.0}, quant_conv / post_quant_conv,
    residual blocks as norm1 / conv1 / norm2 / conv2 / conv_shortcut) and is pinned only structurally: the caller checks that
    the result has exactly the native key set with matching shapes.  Native-named dicts pass through unchanged."""
    if "encoder.conv1.weight" in sd:
        return dict(sd)
    res_map = {"norm1.gamma": "residual.0.gamma", "conv1.": "residual.2.", "norm2.gamma": "residual.3.gamma", "conv2.": "residual.6.",
               "conv_shortcut.": "shortcut."}

    def res(rest: str) -> str:
        for a, b in res_map.items():
            if rest.startswith(a):
                return b + rest[len(a):]
        return rest  # resample.1.* / time_conv.* / norm.gamma / to_qkv.* / proj.* keep their names

    out = {}
    for k, v in sd.items():
        m = re.match(r"^(encoder|decoder)\.(.*)$", k)
        if not m:
            top = {"quant_conv.": "conv1.", "post_quant_conv.": "conv2."}
            for a, b in top.items():
                if k.startswith(a):
                    out[b + k[len(a):]] = v
                    break
            else:
                raise KeyError(f"unrecognised VAE key {k}")
            continue
        side, rest = m.group(1), m.group(2)
        if rest.startswith("conv_in."):
            new = "conv1." + rest[len("conv_in."):]
        elif rest.startswith("conv_out."):
            new = "head.2." + rest[len("conv_out."):]
        elif rest == "norm_out.gamma":
            new = "head.0.gamma"
        elif rest.startswith("mid_block.resnets."):
            j, tail = rest[len("mid_block.resnets."):].split(".", 1)
            new = f"middle.{0 if j == '0' else 2}." + res(tail)
        elif rest.startswith("mid_block.attentions.0."):
            new = "middle.1." + rest[len("mid_block.attentions.0."):]
        elif rest.startswith("down_blocks."):
            j, tail = rest[len("down_blocks."):].split(".", 1)
            new = f"downsamples.{j}." + res(tail)
        elif rest.startswith("up_blocks."):
            i, kind, tail = rest[len("up_blocks."):].split(".", 2)
            base = int(i) * (num_res_blocks + 2)  # num_res_blocks + 1 residual blocks and one upsampler per stage
            if kind == "resnets":
                j, tail2 = tail.split(".", 1)
                new = f"upsamples.{base + int(j)}." + res(tail2)
            elif kind == "upsamplers":
                _, tail2 = tail.split(".", 1)
                new = f"upsamples.{base + num_res_blocks + 1}." + tail2
            else:
                raise KeyError(f"unrecognised VAE key {k}")
        else:
            raise KeyError(f"unrecognised VAE key {k}")
        out[f"{side}.{new}"] = v
    return out


# ------------------------------------------------------------------------------------------
# LoRA
# ------------------------------------------------------------------------------------------
_PREFIXES = ("transformer.", "diffusion_model.", "model.diffusion_model.", "base_model.model.")
_DOWN = (".lora_A.weight", ".lora_down.weight", ".lora_A.default.weight")
_UP = (".lora_B.weight", ".lora_up.weight", ".lora_B.default.weight")


def parse_lora(sd: Dict[str, torch.Tensor]) -> Dict[str, Tuple[torch.Tensor, torch.Tensor, Optional[float]]]:
    """{module path (diffusers name, e.g. ``blocks.0.attn1.to_q``): (A [r, in], B [out, r], alpha or None)} from a
    diffusers / PEFT style LoRA state dict (``<prefix><module>.lora_A.weight`` / ``.lora_B.weight`` / ``.alpha``)."""
    def strip(k: str) -> str:
        for p in _PREFIXES:
            if k.startswith(p):
                return k[len(p):]
        return k

    down, up, alpha = {}, {}, {}
    for k, t in sd.items():
        name = strip(k)
        for suf in _DOWN:
            if name.endswith(suf):
                down[name[: -len(suf)]] = t
                break
        else:
            for suf in _UP:
                if name.endswith(suf):
                    up[name[: -len(suf)]] = t
                    break
            else:
                if name.endswith(".alpha"):
                    alpha[name[: -len(".alpha")]] = float(t)
                else:
                    raise KeyError(f"unrecognised LoRA key {k}")
    if set(down) != set(up):
        odd = sorted(set(down) ^ set(up))
        raise KeyError(f"LoRA A/B halves do not pair up: {odd[:4]}")
    out = {}
    for m in down:
        a, b = down[m], up[m]
        if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[1]:
            raise ValueError(f"{m}: LoRA shapes A {tuple(a.shape)} / B {tuple(b.shape)} do not chain")
        out[m] = (a, b, alpha.get(m))
    return out


class LoraMixin:
    """``load_lora_weights`` / ``fuse_lora`` of the reference pipeline (PEFT through diffusers), on the plain parameter
    tree: adapters are kept on the host until fused; fusing edits the Linear weights in place and drops the packed
    engine so that the next forward re-packs (fused QKV etc.) from the new weights."""

    def _lora_store(self) -> Dict[str, Dict]:
        if not hasattr(self, "_lora_adapters"):
            self._lora_adapters = {}
        return self._lora_adapters

    def load_lora_weights(self, path_or_state, adapter_name: str = "default") -> "LoraMixin":
        if isinstance(path_or_state, (str, os.PathLike)):
            sd = load_state_dict_files([os.fspath(path_or_state)])
        else:
            sd = dict(path_or_state)
        adapter = parse_lora(sd)
        modules = dict(self.named_modules())
        for m, (a, b, _) in adapter.items():
            mod = modules.get(m)
            if not isinstance(mod, torch.nn.Linear):
                raise KeyError(f"LoRA target {m} is not a Linear of this model")
            if a.shape[1] != mod.in_features or b.shape[0] != mod.out_features:
                raise ValueError(f"{m}: LoRA maps {a.shape[1]} -> {b.shape[0]}, the layer {mod.in_features} -> {mod.out_features}")
        store = self._lora_store()
        if adapter_name in store:
            raise ValueError(f"adapter {adapter_name!r} is already loaded")
        store[adapter_name] = adapter
        return self

    @torch.no_grad()
    def fuse_lora(self, adapter_names: Optional[List[str]] = None, lora_scale: float = 1.0) -> "LoraMixin":
        """W += lora_scale * (alpha / r) * B @ A for every target of the named adapters (all loaded ones by default);
        the product is formed in fp32 and rounded once into the weight dtype."""
        store = self._lora_store()
        names = list(store) if adapter_names is None else list(adapter_names)
        modules = dict(self.named_modules())
        fused = getattr(self, "_lora_fused", set())
        for n in names:
            if n not in store:
                raise KeyError(f"no adapter named {n!r} (loaded: {sorted(store)})")
            if n in fused:
                raise ValueError(f"adapter {n!r} is already fused")
            for m, (a, b, alpha) in store[n].items():
                w = modules[m].weight
                r = a.shape[0]
                s = lora_scale * ((alpha / r) if alpha is not None else 1.0)
                delta = b.to(device=w.device, dtype=torch.float32) @ a.to(device=w.device, dtype=torch.float32)
                w.copy_((w.float() + s * delta).to(w.dtype))
            fused.add(n)
        self._lora_fused = fused
        self.invalidate()
        return self

    def unload_lora_weights(self) -> "LoraMixin":
        """Forget adapters that were loaded (fused deltas stay in the weights, as with diffusers after fuse_lora)."""
        self._lora_store().clear()
        return self
