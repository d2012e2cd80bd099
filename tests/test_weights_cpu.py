"""Checkpoint / LoRA plumbing of the drop-in transformer (host side, no GPU): diffusers-layout shards round-trip, the
fp32 islands and the ignored key of the reference class, LoRA parse + fuse against the closed form."""
import json
import os

import pytest
import torch

from chronoedit_amd import weights
from chronoedit_amd.transformer import ChronoEditTransformer3DModel

TINY = dict(num_attention_heads=2, attention_head_dim=128, in_channels=36, out_channels=16, text_dim=64, freq_dim=32, ffn_dim=512,
            num_layers=2, image_dim=48, added_kv_proj_dim=256)


def tiny(seed=0):
    torch.manual_seed(seed)
    m = ChronoEditTransformer3DModel(**TINY)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape).to(p.dtype) * 0.05)
    return m


@pytest.mark.parametrize("shard_bytes", [5 << 30, 1 << 20])
def test_save_load_roundtrip(tmp_path, shard_bytes):
    m = tiny()
    names = m.save_pretrained(str(tmp_path / "transformer"), max_shard_bytes=shard_bytes)
    assert (len(names) > 1) == (shard_bytes == 1 << 20)
    if len(names) > 1:
        idx = json.load(open(tmp_path / "transformer" / weights.INDEX_NAME))
        assert set(idx["weight_map"]) == set(dict(m.named_parameters()))
    m2 = ChronoEditTransformer3DModel.from_pretrained(str(tmp_path), subfolder="transformer", torch_dtype=torch.bfloat16)
    assert vars(m2.config) == vars(m.config)
    for (k, a), (k2, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert k == k2 and a.dtype == b.dtype and torch.equal(a, b), k
    # the fp32 islands of the reference class (transformer_chronoedit.py:338) survive torch_dtype=bf16
    assert m2.scale_shift_table.dtype == torch.float32
    assert m2.blocks[0].scale_shift_table.dtype == torch.float32
    assert m2.condition_embedder.time_embedder.linear_1.weight.dtype == torch.float32
    assert m2.blocks[0].attn1.to_q.weight.dtype == torch.bfloat16


def test_unexpected_and_missing_keys(tmp_path):
    from safetensors.torch import load_file, save_file
    m = tiny()
    d = tmp_path / "t"
    m.save_pretrained(str(d))
    sd = load_file(str(d / weights.WEIGHTS_NAME))
    # norm_added_q is silently ignored like the reference (_keys_to_ignore_on_load_unexpected, :339)
    sd["blocks.0.attn2.norm_added_q.weight"] = torch.ones(256)
    save_file(sd, str(d / weights.WEIGHTS_NAME))
    ChronoEditTransformer3DModel.from_pretrained(str(d))
    sd["blocks.0.attn2.bogus.weight"] = torch.ones(4)
    save_file(sd, str(d / weights.WEIGHTS_NAME))
    with pytest.raises(KeyError, match="unexpected"):
        ChronoEditTransformer3DModel.from_pretrained(str(d))
    del sd["blocks.0.attn2.bogus.weight"], sd["proj_out.bias"]
    save_file(sd, str(d / weights.WEIGHTS_NAME))
    with pytest.raises(KeyError, match="missing"):
        ChronoEditTransformer3DModel.from_pretrained(str(d))
    with pytest.raises(FileNotFoundError):
        ChronoEditTransformer3DModel.from_pretrained(str(tmp_path / "nope"))


def _lora(m, targets, r=4, prefix="transformer.", style=("lora_A", "lora_B"), alpha=None, seed=1):
    g = torch.Generator().manual_seed(seed)
    mods = dict(m.named_modules())
    sd = {}
    for t in targets:
        lin = mods[t]
        sd[f"{prefix}{t}.{style[0]}.weight"] = torch.randn(r, lin.in_features, generator=g) * 0.1
        sd[f"{prefix}{t}.{style[1]}.weight"] = torch.randn(lin.out_features, r, generator=g) * 0.1
        if alpha is not None:
            sd[f"{prefix}{t}.alpha"] = torch.tensor(float(alpha))
    return sd


@pytest.mark.parametrize("prefix,style,alpha", [("transformer.", ("lora_A", "lora_B"), None), ("diffusion_model.", ("lora_down", "lora_up"), 8.0),
                                                ("", ("lora_A", "lora_B"), 2.0)])
def test_lora_fuse_matches_closed_form(tmp_path, prefix, style, alpha):
    from safetensors.torch import save_file
    m = tiny()
    targets = ["blocks.0.attn1.to_q", "blocks.1.attn2.to_out.0", "blocks.1.ffn.net.0.proj", "blocks.0.ffn.net.2"]
    sd = _lora(m, targets, r=4, prefix=prefix, style=style, alpha=alpha)
    before = {t: dict(m.named_modules())[t].weight.detach().clone() for t in targets}
    untouched = m.blocks[0].attn1.to_k.weight.detach().clone()
    m.engine  # noqa: B018  (attribute exists; fuse must drop any packed engine)
    m._engine = object()
    f = tmp_path / "distill.safetensors"
    save_file(sd, str(f))
    m.load_lora_weights(str(f), adapter_name="distill")
    m.fuse_lora(adapter_names=["distill"], lora_scale=0.7)
    assert m._engine is None
    for t in targets:
        a, b = sd[f"{prefix}{t}.{style[0]}.weight"], sd[f"{prefix}{t}.{style[1]}.weight"]
        s = 0.7 * ((alpha / 4) if alpha is not None else 1.0)
        want = (before[t].float() + s * (b @ a)).to(torch.bfloat16)
        assert torch.equal(dict(m.named_modules())[t].weight, want), t
    assert torch.equal(m.blocks[0].attn1.to_k.weight, untouched)
    with pytest.raises(ValueError, match="already fused"):
        m.fuse_lora(adapter_names=["distill"])


def test_two_adapters_and_errors():
    m = tiny()
    a1 = _lora(m, ["blocks.0.attn1.to_q"], seed=1)
    a2 = _lora(m, ["blocks.0.attn1.to_q", "blocks.0.attn1.to_v"], seed=2)
    w0 = m.blocks[0].attn1.to_q.weight.detach().clone()
    m.load_lora_weights(a1, adapter_name="a").load_lora_weights(a2, adapter_name="b")
    with pytest.raises(ValueError, match="already loaded"):
        m.load_lora_weights(a1, adapter_name="a")
    m.fuse_lora(lora_scale=1.0)  # both, in load order; each rounds into bf16 once
    k = "transformer.blocks.0.attn1.to_q."
    step1 = (w0.float() + a1[k + "lora_B.weight"] @ a1[k + "lora_A.weight"]).to(torch.bfloat16)
    step2 = (step1.float() + a2[k + "lora_B.weight"] @ a2[k + "lora_A.weight"]).to(torch.bfloat16)
    assert torch.equal(m.blocks[0].attn1.to_q.weight, step2)
    with pytest.raises(KeyError, match="not a Linear"):
        tiny().load_lora_weights({"transformer.blocks.0.norm2.lora_A.weight": torch.zeros(2, 256),
                                  "transformer.blocks.0.norm2.lora_B.weight": torch.zeros(256, 2)})
    with pytest.raises(ValueError, match="LoRA maps"):
        tiny().load_lora_weights({"transformer.blocks.0.attn1.to_q.lora_A.weight": torch.zeros(2, 100),
                                  "transformer.blocks.0.attn1.to_q.lora_B.weight": torch.zeros(256, 2)})
    with pytest.raises(KeyError, match="pair up"):
        weights.parse_lora({"transformer.blocks.0.attn1.to_q.lora_A.weight": torch.zeros(2, 256)})
    with pytest.raises(KeyError, match="unrecognised"):
        weights.parse_lora({"transformer.blocks.0.attn1.to_q.weight": torch.zeros(2, 256)})
    with pytest.raises(KeyError, match="no adapter"):
        tiny().fuse_lora(adapter_names=["zzz"])


def test_pipeline_delegates_lora():
    from chronoedit_amd.pipeline import ChronoEditPipeline
    m = tiny()
    pipe = ChronoEditPipeline(vae=None, transformer=m, scheduler=None)
    a = _lora(m, ["blocks.1.attn1.to_k"])
    w0 = m.blocks[1].attn1.to_k.weight.detach().clone()
    pipe.load_lora_weights(a, adapter_name="x")
    pipe.fuse_lora(adapter_names=["x"], lora_scale=0.5)
    assert not torch.equal(m.blocks[1].attn1.to_k.weight, w0)


def test_wan_native_names_match_the_reference_converter(golden_dir):
    """diffusers <-> Wan-native key map == what the reference's own WanModelStateDictConverter.from_diffusers produces
    (tests/golden/wan_native_keymap.json, generated by oracle/gen_golden_keymap.py from wan_video_dit_chronoedit.py:434-505)."""
    fx = json.load(open(os.path.join(golden_dir, "wan_native_keymap.json")))
    assert not fx["dropped_by_reference_converter"]
    for d, n in fx["pairs"]:
        assert weights.diffusers_to_wan_native_key(d) == n, (d, n)
    with pytest.raises(KeyError):
        weights.diffusers_to_wan_native_key("blocks.0.attn9.to_q.weight")


def test_wan_native_checkpoint_roundtrip():
    m = tiny(seed=3)
    native = m.wan_native_state_dict()
    assert "blocks.1.cross_attn.k_img.weight" in native and "head.modulation" in native and "blocks.0.modulation" in native
    assert not any(k.startswith(("condition_embedder", "proj_out")) or ".attn1." in k for k in native)
    m2 = tiny(seed=4)
    m2._engine = object()
    m2.load_wan_native_state_dict({k: v.clone() for k, v in native.items()})
    assert m2._engine is None
    for (k, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), k
    with pytest.raises(KeyError):
        m2.load_wan_native_state_dict({**native, "blocks.0.bogus.weight": torch.zeros(1)})


def _diffusers_vae_names(native_keys, num_res_blocks=2):
    """Independent forward map native -> diffusers layout (test-side), to exercise the product's inverse."""
    res = {"residual.0.gamma": "norm1.gamma", "residual.2.": "conv1.", "residual.3.gamma": "norm2.gamma", "residual.6.": "conv2.",
           "shortcut.": "conv_shortcut."}
    out = {}
    for k in native_keys:
        side, rest = (k.split(".", 1) + [""])[:2] if k.split(".")[0] in ("encoder", "decoder") else (None, k)
        if side is None:
            out[k] = ("quant_conv." if k.startswith("conv1.") else "post_quant_conv.") + k.split(".", 1)[1]
            continue
        def r(t):
            for a, b in res.items():
                if t.startswith(a):
                    return b + t[len(a):]
            return t
        if rest.startswith("conv1."):
            new = "conv_in." + rest[6:]
        elif rest.startswith("head.2."):
            new = "conv_out." + rest[7:]
        elif rest == "head.0.gamma":
            new = "norm_out.gamma"
        elif rest.startswith("middle."):
            j, t = rest[7:].split(".", 1)
            new = f"mid_block.attentions.0.{t}" if j == "1" else f"mid_block.resnets.{0 if j == '0' else 1}.{r(t)}"
        elif rest.startswith("downsamples."):
            j, t = rest[12:].split(".", 1)
            new = f"down_blocks.{j}.{r(t)}"
        else:
            j, t = rest[len("upsamples."):].split(".", 1)
            i, jj = divmod(int(j), num_res_blocks + 2)
            new = f"up_blocks.{i}.upsamplers.0.{t}" if jj == num_res_blocks + 1 else f"up_blocks.{i}.resnets.{jj}.{r(t)}"
        out[k] = f"{side}.{new}"
    return out


def test_vae_checkpoint_naming_roundtrip(tmp_path):
    """AutoencoderKLWan.from_pretrained accepts the native Wan names and the diffusers layout (structural check only: diffusers is
    not installed here, see wan_vae_diffusers_to_native)."""
    from safetensors.torch import save_file
    from chronoedit_amd.vae import wan_vae_param_shapes
    arch = dict(dim=32, z_dim=16)
    shapes = wan_vae_param_shapes(**arch)
    g = torch.Generator().manual_seed(0)
    native = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    fwd = _diffusers_vae_names(shapes)
    assert len(set(fwd.values())) == len(fwd)
    assert "decoder.up_blocks.3.resnets.2.conv2.weight" in fwd.values() and "encoder.mid_block.attentions.0.to_qkv.weight" in fwd.values()
    diff = {fwd[k]: v for k, v in native.items()}
    back = weights.wan_vae_diffusers_to_native(diff)
    assert set(back) == set(native) and all(torch.equal(back[k], native[k]) for k in native)
    assert weights.wan_vae_diffusers_to_native(native).keys() == native.keys()  # native passes through
    with pytest.raises(KeyError):
        weights.wan_vae_diffusers_to_native({**diff, "encoder.bogus.weight": torch.zeros(1)})
    # from_pretrained end to end on CPU tensors (device="cpu": only the loader is exercised, no kernels run)
    from chronoedit_amd.vae import AutoencoderKLWan
    d = tmp_path / "vae"
    d.mkdir()
    json.dump({"base_dim": 32, "z_dim": 16, "dim_mult": [1, 2, 4, 4], "num_res_blocks": 2, "temperal_downsample": [False, True, True],
               "_class_name": "AutoencoderKLWan"}, open(d / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in diff.items()}, str(d / weights.WEIGHTS_NAME))
    vae = AutoencoderKLWan.from_pretrained(str(tmp_path), subfolder="vae", device="cpu")
    assert vae.config.z_dim == 16 and torch.equal(vae._params["decoder.head.2.weight"], native["decoder.head.2.weight"])


def test_pipeline_from_pretrained_directory_layout(tmp_path):
    """The diffusers model-directory layout the reference's runner points at: transformer/, vae/, text_encoder/, image_encoder/,
    scheduler/ - every drop-in loads itself from its subfolder (CPU tensors here: loaders only, no kernels)."""
    from safetensors.torch import save_file
    from chronoedit_amd.clip_vision import CLIPVisionModel
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.umt5 import UMT5EncoderModel
    from chronoedit_amd.vae import wan_vae_param_shapes
    root = tmp_path / "ChronoEdit-tiny"
    t = tiny(seed=5)
    t.save_pretrained(str(root / "transformer"))
    (root / "vae").mkdir()
    json.dump({"base_dim": 32, "z_dim": 16}, open(root / "vae" / "config.json", "w"))
    g = torch.Generator().manual_seed(1)
    save_file({k: torch.randn(s, generator=g) for k, s in wan_vae_param_shapes(dim=32, z_dim=16).items()}, str(root / "vae" / weights.WEIGHTS_NAME))
    te = UMT5EncoderModel(vocab_size=50, d_model=128, d_kv=64, d_ff=256, num_layers=1, num_heads=2, device="cpu")
    (root / "text_encoder").mkdir()
    json.dump({**vars(te.config), "model_type": "umt5"}, open(root / "text_encoder" / "config.json", "w"))
    sd = {k: v.detach().clone() for k, v in te.state_dict().items()}
    save_file(sd, str(root / "text_encoder" / "model.safetensors"))
    ie = CLIPVisionModel(hidden_size=320, intermediate_size=640, num_hidden_layers=1, num_attention_heads=4, image_size=28, patch_size=14, device="cpu")
    (root / "image_encoder").mkdir()
    json.dump(vars(ie.config), open(root / "image_encoder" / "config.json", "w"))
    save_file({k: v.detach().clone() for k, v in ie.state_dict().items()}, str(root / "image_encoder" / "model.safetensors"))
    (root / "scheduler").mkdir()
    json.dump({"_class_name": "UniPCMultistepScheduler", "flow_shift": 3.0, "solver_order": 2, "use_flow_sigmas": True,
               "prediction_type": "flow_prediction"}, open(root / "scheduler" / "scheduler_config.json", "w"))
    pipe = ChronoEditPipeline.from_pretrained(str(root), device="cpu")
    assert pipe.scheduler.config.shift == 3.0
    for (k, a), (_, b) in zip(t.named_parameters(), pipe.transformer.named_parameters()):
        assert torch.equal(a, b), k
    assert torch.equal(pipe.text_encoder.shared.weight, te.shared.weight)
    assert pipe.text_encoder.encoder.embed_tokens.weight is pipe.text_encoder.shared.weight  # tied, as in transformers
    assert torch.equal(pipe.image_encoder.vision_model.embeddings.class_embedding, ie.vision_model.embeddings.class_embedding)
    assert pipe.vae.config.z_dim == 16
    pipe2 = ChronoEditPipeline.from_pretrained(str(root), transformer=t, device="cpu", load_encoders=False)
    assert pipe2.transformer is t and pipe2.text_encoder is None
