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
