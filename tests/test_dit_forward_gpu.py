"""End-to-end DiT forward parity on the GPU: the HIP engine behind the reference's
ChronoEditTransformer3DModel.forward signature vs
  (a) the committed golden vectors produced by the reference's own transformer file
      (tests/golden/dit_*_bf16.pt, oracle/gen_golden.py), and
  (b) the CPU oracle run here in fp32 and bf16 on the same seeded inputs.
Tolerance: bf16 HIP vs fp32 oracle rel-L2 <= 2e-2 per forward, and no worse than
3x the bf16 eager oracle's own error vs fp32 (both printed)."""
import os

import pytest
import torch

from oracle import device as OD
from oracle import dit_oracle as O

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _build(cfg: O.DiTConfig, params):
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    m = ChronoEditTransformer3DModel(
        num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim, in_channels=cfg.in_channels,
        out_channels=cfg.out_channels, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim, ffn_dim=cfg.ffn_dim,
        num_layers=cfg.num_layers, image_dim=cfg.image_dim, added_kv_proj_dim=cfg.added_kv_proj_dim,
        rope_temporal_skip_len=cfg.rope_temporal_skip_len, device="cuda:0", dtype=torch.bfloat16)
    m.load_synthetic_({k: v.to("cuda:0") for k, v in params.items()})
    return m


@pytest.mark.parametrize("name", ["tiny_T2_bf16", "tiny_T8_bf16", "small_T2_bf16"])
def test_forward_matches_reference_golden(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, f"dit_{name}.pt"))
    cfg = O.DiTConfig(**fx["cfg"])
    p_bf = O.make_synthetic_params(cfg, seed=fx["param_seed"], dtype=torch.bfloat16)
    lat, text, image = O.make_synthetic_inputs(cfg, fx["T"], fx["h"], fx["w"], dtype=torch.bfloat16,
                                               text_len=fx["text_len"], real_text=fx["real_text"])
    model = _build(cfg, p_bf)
    ts = torch.tensor([fx["timestep"]], device="cuda:0")
    out = model(lat.cuda(), ts, text.cuda(), image.cuda(), return_dict=False)[0]
    assert out.shape == fx["out"].shape and out.dtype == torch.bfloat16
    # fp32 oracle on fp32 copies of the same (bf16-representable) weights/inputs
    p32 = {k: v.float() for k, v in p_bf.items()}
    with torch.no_grad():
        ref32 = O.dit_forward(p32, cfg, lat.float(), torch.tensor([fx["timestep"]]), text.float(), image.float())
    e_hip = rel_l2(out, ref32)
    e_eager = rel_l2(fx["out"], ref32)
    e_vs_golden = rel_l2(out, fx["out"])
    print(f"{name}: hip-vs-fp32 {e_hip:.3e}  ref-bf16-eager-vs-fp32 {e_eager:.3e}  hip-vs-golden {e_vs_golden:.3e}")
    assert e_hip <= 2e-2
    assert e_hip <= 3 * e_eager + 2e-3
    assert e_vs_golden <= 2e-2


def test_forward_signature_and_errors():
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=1, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    model = _build(cfg, O.make_synthetic_params(cfg, dtype=torch.bfloat16))
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 8, 8, dtype=torch.bfloat16, text_len=32, real_text=8)
    ts = torch.tensor([10], device="cuda:0")
    out = model(lat.cuda(), ts, text.cuda(), image.cuda())
    assert hasattr(out, "sample") and out.sample.shape == (1, 16, 2, 8, 8)
    with pytest.raises(AssertionError):  # transformer_chronoedit.py:205
        bad = torch.zeros(1, 36, 3, 8, 8, dtype=torch.bfloat16, device="cuda:0")
        model(bad, ts, text.cuda(), image.cuda())
    from chronoedit_amd import ops
    with pytest.raises(ops.HipKernelError):  # no CPU fallback
        model(lat, ts.cpu(), text, image)
    # context cache returns identical results
    model.cache_context = True
    a = model(lat.cuda(), ts, text.cuda(), image.cuda()).sample
    b = model(lat.cuda(), ts, text.cuda(), image.cuda()).sample
    assert torch.equal(a, b) and torch.equal(a, out.sample)
    # ... also after an UNCACHED call with another conditioning rewrote the engine-owned V^T buffers the cached entry had views into
    # (the stale entry used to pair the old K with the other conditioning's V^T)
    tx, im = text.cuda(), image.cuda()
    a = model(lat.cuda(), ts, tx, im).sample  # cached under (tx, im)
    model.cache_context = False
    other = torch.randn(1, 32, 128, generator=torch.Generator().manual_seed(5)).to(torch.bfloat16).cuda()
    c = model(lat.cuda(), ts, other, (im * -1.5).contiguous()).sample
    assert not torch.equal(c, a)
    model.cache_context = True
    b = model(lat.cuda(), ts, tx, im).sample
    assert torch.equal(a, b)


def test_batched_forward_equals_sequential():
    """B=2 (the batched classifier-free-guidance form) == two B=1 forwards, bit for bit with the same GEMM kernel."""
    from chronoedit_amd import ops
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    model = _build(cfg, O.make_synthetic_params(cfg, dtype=torch.bfloat16))
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 16, dtype=torch.bfloat16, text_len=48, real_text=8)
    text_b = torch.randn(1, 48, 128, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16)
    ts = torch.tensor([500], device="cuda:0")
    old = ops.set_gemm_variant(0)
    try:
        a = model(lat.cuda(), ts, text.cuda(), image.cuda()).sample
        b = model(lat.cuda(), ts, text_b.cuda(), image.cuda()).sample
        both = model(torch.cat([lat, lat]).cuda(), torch.cat([ts, ts]), torch.cat([text, text_b]).cuda(),
                     torch.cat([image, image]).cuda()).sample
    finally:
        ops.set_gemm_variant(old)
    assert torch.equal(both[0], a[0]) and torch.equal(both[1], b[0])
    assert not torch.equal(a, b)


def test_full_width_block_at_720p_vs_fp32_reference():
    """BASELINE configs[1] shapes (D = 5120, 40 heads, F = 13824, N = 7200 tokens, 512 + 257 context rows) on ONE block:
    the 256-tile LDS-DMA GEMM, the ping-pong attention at 7200 keys x 40 heads and the 14B row kernels, against the fp32
    CPU oracle (~20 s on the GPU box's host cores)."""
    cfg = O.DiTConfig(num_layers=1)
    p_bf = O.make_synthetic_params(cfg, seed=7, dtype=torch.bfloat16)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 90, 160, dtype=torch.bfloat16)
    model = _build(cfg, p_bf)
    ts = torch.tensor([800], device="cuda:0")
    out = model(lat.cuda(), ts, text.cuda(), image.cuda(), return_dict=False)[0]
    with torch.no_grad(), OD.on() as dev:  # the fp32 oracle, evaluated where oracle/device.py says (CE_ORACLE_DEVICE=cpu: the host cores, ~20 s)
        ref = O.dit_forward(OD.to(p_bf, dev, torch.float32), cfg, OD.to(lat, dev, torch.float32), torch.tensor([800]), OD.to(text, dev, torch.float32),
                            OD.to(image, dev, torch.float32)).cpu()
    e = rel_l2(out, ref)
    print(f"full-width block @N=7200: rel-L2 vs fp32 {e:.3e}")
    assert out.shape == (1, 16, 2, 90, 160) and torch.isfinite(out.float()).all()
    assert e < 1e-2


def test_lora_fuse_and_checkpoint_roundtrip_drive_the_engine(tmp_path):
    """Weights edited through the reference's entry points (fuse_lora, from_pretrained) reach the packed HIP operands:
    the forward of the fused model == the forward of a fresh model loaded from its saved checkpoint, and both move away
    from the un-fused output by what the fp32 oracle predicts for the same weight edit."""
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p_bf = O.make_synthetic_params(cfg, dtype=torch.bfloat16)
    model = _build(cfg, p_bf)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 16, dtype=torch.bfloat16)
    ts = torch.tensor([500], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    base = model(*args, return_dict=False)[0].clone()
    g = torch.Generator().manual_seed(5)
    lora = {}
    for t in ["blocks.0.attn1.to_q", "blocks.0.attn1.to_v", "blocks.1.ffn.net.0.proj", "blocks.1.attn2.to_out.0"]:
        lin = dict(model.named_modules())[t]
        lora[f"transformer.{t}.lora_A.weight"] = torch.randn(8, lin.in_features, generator=g) * 0.05
        lora[f"transformer.{t}.lora_B.weight"] = torch.randn(lin.out_features, 8, generator=g) * 0.05
    model.load_lora_weights(lora, adapter_name="distill")
    model.fuse_lora(adapter_names=["distill"], lora_scale=1.0)
    fused = model(*args, return_dict=False)[0].clone()
    assert rel_l2(fused, base) > 1e-3  # the edit is visible
    model.save_pretrained(str(tmp_path / "transformer"), max_shard_bytes=1 << 20)
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    again = ChronoEditTransformer3DModel.from_pretrained(str(tmp_path), subfolder="transformer", torch_dtype=torch.bfloat16, device="cuda:0")
    assert torch.equal(again(*args, return_dict=False)[0], fused)
    # oracle with the fused (bf16-rounded) weights, fp32 arithmetic
    p32 = {k: v.detach().float().cpu() for k, v in model.named_parameters()}
    with torch.no_grad():
        ref = O.dit_forward(p32, cfg, lat.float(), torch.tensor([500]), text.float(), image.float())
    assert rel_l2(fused, ref) <= 2e-2


def test_full_depth_forty_blocks_error_growth():
    """Depth of the real model (40 blocks) at a narrow width: the bf16 error against the fp32 oracle accumulates over the
    residual stream but stays inside the per-forward tolerance, and no worse than bf16-eager arithmetic (the oracle in bf16)."""
    cfg = O.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=40, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    p_bf = O.make_synthetic_params(cfg, seed=11, dtype=torch.bfloat16)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, 16, 24, dtype=torch.bfloat16)
    model = _build(cfg, p_bf)
    ts = torch.tensor([700], device="cuda:0")
    out = model(lat.cuda(), ts, text.cuda(), image.cuda(), return_dict=False)[0]
    p32 = {k: v.float() for k, v in p_bf.items()}
    with torch.no_grad():
        ref32 = O.dit_forward(p32, cfg, lat.float(), torch.tensor([700]), text.float(), image.float())
        ref_bf = O.dit_forward(p_bf, cfg, lat, torch.tensor([700]), text, image)
    e_hip, e_eager = rel_l2(out, ref32), rel_l2(ref_bf, ref32)
    print(f"40 blocks: hip-vs-fp32 {e_hip:.3e}  bf16-eager-oracle-vs-fp32 {e_eager:.3e}")
    assert torch.isfinite(out.float()).all()
    assert e_hip <= 2e-2 and e_hip <= 3 * e_eager + 2e-3
