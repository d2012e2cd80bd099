"""Conditioning encoders on the GPU: the HIP drop-ins behind the transformers call signatures vs the golden vectors of the
real transformers classes and the fp32 CPU oracles.  Tolerance: bf16 HIP vs fp32 oracle rel-L2 <= 2e-2 and no worse than
3x the reference's own bf16-eager error (both printed)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _clip(cfg, params):
    from chronoedit_amd.clip_vision import CLIPVisionModel
    m = CLIPVisionModel(**vars(cfg), device="cuda:0", dtype=torch.bfloat16)
    own = dict(m.named_parameters())
    assert set(own) == set(params), (sorted(set(own) ^ set(params))[:6])
    with torch.no_grad():
        for k, p in own.items():
            p.copy_(params[k].to(p.dtype))
    m.invalidate()
    return m


def test_clip_vision_matches_transformers_golden(golden_dir):
    from oracle import clip_oracle as C
    fx = torch.load(os.path.join(golden_dir, "clip_tiny.pt"))
    cfg = C.CLIPVisionCfg(**fx["cfg"])
    p = C.make_synthetic_params(cfg, fx["param_seed"])
    px = C.make_synthetic_pixels(cfg, fx["batch"], fx["pixel_seed"])
    model = _clip(cfg, {k: v.to(torch.bfloat16) for k, v in p.items()})
    out = model(pixel_values=px.cuda(), output_hidden_states=True)
    assert len(out.hidden_states) == cfg.num_hidden_layers + 1
    got = out.hidden_states[-2]
    assert got.shape == fx["penultimate_fp32"].shape and got.dtype == torch.bfloat16
    # fp32 oracle on the bf16-rounded weights (what both bf16 implementations actually hold)
    p32 = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    ref32 = C.clip_vision_hidden_states(p32, cfg, px.to(torch.bfloat16).float())[-2]
    e_hip, e_eager = rel_l2(got, ref32), rel_l2(fx["penultimate_bf16"], ref32)
    print(f"clip tiny: hip-vs-fp32 {e_hip:.3e}  transformers-bf16-vs-fp32 {e_eager:.3e}  hip-vs-golden {rel_l2(got, fx['penultimate_bf16']):.3e}")
    assert e_hip <= 2e-2 and e_hip <= 3 * e_eager + 2e-3
    assert torch.equal(out.last_hidden_state, out.hidden_states[-1])
    assert out.pooler_output.shape == (fx["batch"], cfg.hidden_size)
    # batch of 2 in one call == two calls of 1 (the stacked-sample attention launch)
    one = model(pixel_values=px[1:].cuda(), output_hidden_states=True).hidden_states[-2]
    assert torch.equal(one[0], got[1])
    with pytest.raises(Exception):
        model(pixel_values=px)  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        model(pixel_values=torch.zeros(1, 3, 28, 28, device="cuda:0"))


def test_clip_vision_vith_shape_runs_and_is_finite():
    """One full-width ViT-H/14 layer stack slice (width 1280, 16 heads x 80, 257 tokens) vs the fp32 oracle."""
    from oracle import clip_oracle as C
    cfg = C.CLIPVisionCfg(num_hidden_layers=2)
    p = C.make_synthetic_params(cfg, seed=5)
    px = C.make_synthetic_pixels(cfg, 1, 3)
    model = _clip(cfg, {k: v.to(torch.bfloat16) for k, v in p.items()})
    got = model(pixel_values=px.cuda(), output_hidden_states=True).hidden_states[-2]
    assert got.shape == (1, 257, 1280) and torch.isfinite(got.float()).all()
    p32 = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    ref32 = C.clip_vision_hidden_states(p32, cfg, px.to(torch.bfloat16).float())[-2]
    e = rel_l2(got, ref32)
    print(f"clip ViT-H width, 2 layers: hip-vs-fp32 {e:.3e}")
    assert e <= 2e-2


def _umt5(cfg, params):
    from chronoedit_amd.umt5 import UMT5EncoderModel
    m = UMT5EncoderModel(**vars(cfg), device="cuda:0", dtype=torch.bfloat16)
    own = dict(m.named_parameters())
    assert set(own) == set(params), (sorted(set(own) ^ set(params))[:6])
    with torch.no_grad():
        for k, p in own.items():
            p.copy_(params[k].to(p.dtype))
    m.invalidate()
    return m


def test_umt5_encoder_matches_transformers_golden(golden_dir):
    from chronoedit_amd.umt5 import t5_prompt_embeds
    from oracle import umt5_oracle as U
    fx = torch.load(os.path.join(golden_dir, "umt5_tiny.pt"))
    cfg = U.UMT5Cfg(**fx["cfg"])
    p = U.make_synthetic_params(cfg, fx["param_seed"])
    ids, mask = U.make_synthetic_tokens(cfg, fx["lens"], fx["L"], fx["token_seed"])
    model = _umt5(cfg, {k: v.to(torch.bfloat16) for k, v in p.items()})
    got = model(ids.cuda(), mask.cuda()).last_hidden_state
    assert got.shape == fx["last_fp32"].shape and got.dtype == torch.bfloat16
    p32 = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    ref32 = U.umt5_encode(p32, cfg, ids, mask)
    e_hip, e_eager = rel_l2(got, ref32), rel_l2(fx["last_bf16"], ref32)
    print(f"umt5 tiny: hip-vs-fp32 {e_hip:.3e} (all rows)  transformers-bf16-vs-fp32 {e_eager:.3e}  hip-vs-golden {rel_l2(got, fx['last_bf16']):.3e}")
    assert e_hip <= 2e-2 and e_hip <= 3 * e_eager + 2e-3
    pe = t5_prompt_embeds(model, ids.cuda(), mask.cuda())
    assert pe[0, fx["lens"][0]:].abs().max().item() == 0 and torch.equal(pe[1], got[1])
    assert rel_l2(pe, U.prompt_embeds(ref32, mask)) <= 2e-2
    with pytest.raises(Exception):
        model(ids, mask)  # CPU tensors: no fallback
    bad = mask.clone()
    bad[0, 3] = 0
    with pytest.raises(ValueError):
        model(ids.cuda(), bad.cuda())


def test_umt5_full_width_layer_at_512_tokens():
    """UMT5-XXL width (d_model 4096, 64 heads x 64, d_ff 10240), one block, two prompts padded to 512 tokens vs the fp32 oracle."""
    from oracle import umt5_oracle as U
    cfg = U.UMT5Cfg(vocab_size=512, num_layers=1)
    p = U.make_synthetic_params(cfg, seed=9)
    ids, mask = U.make_synthetic_tokens(cfg, lens=[40, 333], L=512, seed=2)
    model = _umt5(cfg, {k: v.to(torch.bfloat16) for k, v in p.items()})
    got = model(ids.cuda(), mask.cuda()).last_hidden_state
    assert got.shape == (2, 512, 4096) and torch.isfinite(got.float()).all()
    p32 = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    ref32 = U.umt5_encode(p32, cfg, ids, mask)
    valid = mask.bool()
    e = rel_l2(got.cpu()[valid], ref32[valid])
    print(f"umt5 XXL width, 1 block, 2 x 512 tokens: hip-vs-fp32 {e:.3e} (valid rows)")
    assert e <= 2e-2
