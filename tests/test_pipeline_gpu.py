"""One whole edit (VAE encode -> 4-step CFG denoise -> VAE decode) on the HIP engine vs the fp32 CPU oracle pipeline —
BASELINE.json configs[0] in miniature (small widths, full structure).  Tolerance: final latents rel-L2 <= 6e-2 after 4
steps x 2 forwards x 2 blocks in bf16, decoded video <= 8e-2; also checked: batched CFG == sequential CFG."""
import pytest
import torch

from oracle import device as OD
from oracle import dit_oracle as D
from oracle import pipeline_oracle as P
from oracle import vae_oracle as V

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_edit_end_to_end_vs_oracle():
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from chronoedit_amd.vae import AutoencoderKLWan
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    vcfg = V.VAEConfig(dim=32, z_dim=16)
    dp = D.make_synthetic_params(dcfg, dtype=torch.bfloat16)
    vp = V.make_synthetic_params(vcfg)
    g = torch.Generator().manual_seed(0)
    H, W, F = 64, 96, 5
    image = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    prompt = torch.randn(1, 40, 128, generator=g)
    negative = torch.randn(1, 40, 128, generator=g)
    img_emb = torch.randn(1, 257, 64, generator=g)
    lat0 = torch.randn(1, 16, 2, H // 8, W // 8, generator=g)
    bf = torch.bfloat16
    # fp32 oracle on the bf16-representable weights / inputs
    dp32 = {k: v.float() for k, v in dp.items()}
    with torch.no_grad():
        lat_ref, vid_ref = P.edit(dp32, dcfg, vp, vcfg, image.to(bf).float(), prompt.to(bf).float(), negative.to(bf).float(),
                                  img_emb.to(bf).float(), lat0.clone(), num_frames=F, steps=4)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    vae = AutoencoderKLWan({k: v.cuda() for k, v in vp.items()}, dim=32, z_dim=16)
    pipe = ChronoEditPipeline(vae=vae, transformer=m, scheduler=FlowUniPCMultistepScheduler(flow_shift=5.0))
    args = (image.cuda().to(bf), prompt.cuda().to(bf), negative.cuda().to(bf), img_emb.cuda().to(bf))
    lat = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), output_type="latent")
    vid = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda())
    e_lat, e_vid = rel_l2(lat, lat_ref), rel_l2(vid, vid_ref)
    print(f"edit: latents rel-L2 {e_lat:.3e}, video rel-L2 {e_vid:.3e}")
    assert vid.shape == (1, 3, F, H, W)
    assert e_lat < 6e-2 and e_vid < 8e-2


def test_hipgraph_replay_equals_eager_loop():
    """The hipGraph-captured step replayed over the schedule reproduces the eager loop bit for bit, including the
    temporal-reasoning truncation (8 -> 2 latent frames, second graph)."""
    from chronoedit_amd.pipeline import denoise
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    dp = D.make_synthetic_params(dcfg, dtype=torch.bfloat16)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    g = torch.Generator().manual_seed(1)
    bf = torch.bfloat16
    lat0 = torch.randn(1, 16, 8, 8, 12, generator=g).cuda()
    cond = torch.randn(1, 20, 8, 8, 12, generator=g).cuda().to(bf)
    prompt = torch.randn(1, 40, 128, generator=g).cuda().to(bf)
    negative = torch.randn(1, 40, 128, generator=g).cuda().to(bf)
    img = torch.randn(1, 257, 64, generator=g).cuda().to(bf)
    outs = []
    for use_graph in (False, True):
        sch = FlowUniPCMultistepScheduler(flow_shift=5.0)
        outs.append(denoise(m, sch, lat0.clone(), cond, prompt, negative, img, 6, 5.0, enable_temporal_reasoning=True,
                            num_temporal_reasoning_steps=3, use_graph=use_graph).clone())
    assert outs[0].shape == (1, 16, 2, 8, 12)
    assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max()


def test_pipeline_encoders_feed_the_edit():
    """encode_prompt / encode_image (HIP UMT5 + CLIP drop-ins) produce the conditioning tensors the edit consumes: shapes and
    zero padding as pipeline_chronoedit.py:231-256, and a 2-step edit on them runs finite."""
    from chronoedit_amd.clip_vision import CLIPVisionModel
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.umt5 import UMT5EncoderModel
    torch.manual_seed(0)
    te = UMT5EncoderModel(vocab_size=300, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, device="cuda:0")
    ie = CLIPVisionModel(hidden_size=320, intermediate_size=640, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14,
                         device="cuda:0")
    pipe = ChronoEditPipeline(vae=None, transformer=None, scheduler=None, text_encoder=te, image_encoder=ie)
    ids = torch.randint(2, 300, (1, 64), device="cuda:0")
    am = torch.zeros((1, 64), dtype=torch.long, device="cuda:0")
    am[0, :11] = 1
    nids, nam = torch.randint(2, 300, (1, 64), device="cuda:0"), torch.zeros((1, 64), dtype=torch.long, device="cuda:0")
    nam[0, :5] = 1
    pos, neg = pipe.encode_prompt(input_ids=ids, attention_mask=am, negative_input_ids=nids, negative_attention_mask=nam)
    assert pos.shape == neg.shape == (1, 64, 128) and pos.dtype == torch.bfloat16
    assert pos[0, 11:].abs().max().item() == 0 and neg[0, 5:].abs().max().item() == 0 and pos[0, :11].abs().max().item() > 0
    img = pipe.encode_image(torch.randn(1, 3, 56, 56, device="cuda:0"))
    assert img.shape == (1, 17, 320) and torch.isfinite(img.float()).all()
    with pytest.raises(ValueError):
        ChronoEditPipeline().encode_image(torch.zeros(1, 3, 56, 56, device="cuda:0"))


def test_temporal_reasoning_edit_vs_oracle():
    """The reasoning mode of the reference (pipeline_chronoedit.py:700-709, 776-779): 29 pixel frames = 8 latent frames for the first
    steps, truncated to the first and last latent frame (with the scheduler history) afterwards, decoded as reasoning video +
    edited frame.  HIP pipeline vs the fp32 oracle."""
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from chronoedit_amd.vae import AutoencoderKLWan
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    vcfg = V.VAEConfig(dim=32, z_dim=16)
    dp = D.make_synthetic_params(dcfg, dtype=torch.bfloat16)
    vp = V.make_synthetic_params(vcfg)
    g = torch.Generator().manual_seed(1)
    H, W, F = 64, 96, 29
    image = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    prompt = torch.randn(1, 40, 128, generator=g)
    negative = torch.randn(1, 40, 128, generator=g)
    img_emb = torch.randn(1, 257, 64, generator=g)
    lat0 = torch.randn(1, 16, 8, H // 8, W // 8, generator=g)
    bf = torch.bfloat16
    kw = dict(enable_temporal_reasoning=True, num_temporal_reasoning_steps=2)
    with torch.no_grad(), OD.on() as dev:  # the fp32 oracle (29-frame VAE included) where oracle/device.py says; CE_ORACLE_DEVICE=cpu: the host cores
        dp32, vp32 = OD.to(dp, dev, torch.float32), OD.to(vp, dev)
        oargs = tuple(OD.to(x.to(bf).float(), dev) for x in (image, prompt, negative, img_emb))
        lat_ref, vid_ref = (x.cpu() for x in P.edit(dp32, dcfg, vp32, vcfg, *oargs, lat0.clone().to(dev), num_frames=F, steps=4, **kw))
    assert lat_ref.shape[2] == 2 and vid_ref.shape[2] == 5
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    vae = AutoencoderKLWan({k: v.cuda() for k, v in vp.items()}, dim=32, z_dim=16)
    pipe = ChronoEditPipeline(vae=vae, transformer=m, scheduler=FlowUniPCMultistepScheduler(flow_shift=5.0))
    args = (image.cuda().to(bf), prompt.cuda().to(bf), negative.cuda().to(bf), img_emb.cuda().to(bf))
    lat = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), output_type="latent", **kw)
    vid = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), **kw)
    e_lat, e_vid = rel_l2(lat, lat_ref), rel_l2(vid, vid_ref)
    print(f"reasoning edit: latents rel-L2 {e_lat:.3e}, video rel-L2 {e_vid:.3e}")
    assert lat.shape == lat_ref.shape and vid.shape == vid_ref.shape
    assert e_lat < 6e-2 and e_vid < 8e-2
    # never truncated (k >= steps): 8 latent frames to the end, reasoning video (all but the last latent frame) + edited frame
    kw2 = dict(enable_temporal_reasoning=True, num_temporal_reasoning_steps=50)
    with torch.no_grad(), OD.on() as dev:
        lat_ref2, vid_ref2 = (x.cpu() for x in P.edit(dp32, dcfg, vp32, vcfg, *oargs, lat0.clone().to(dev), num_frames=F, steps=3, **kw2))
    vid2 = pipe.edit_tensors(*args, num_frames=F, num_inference_steps=3, guidance_scale=5.0, latents=lat0.cuda(), **kw2)
    print(f"reasoning (no truncation): video rel-L2 {rel_l2(vid2, vid_ref2):.3e}  shape {tuple(vid2.shape)}")
    assert vid2.shape == vid_ref2.shape and rel_l2(vid2, vid_ref2) < 8e-2


def _oracle_loop(dp, dcfg, lat0, cond, prompt, negative, img, steps, guidance, dtype):
    """The reference loop (pipeline_chronoedit.py:694-756) on the CPU oracles, everything in `dtype`: float32 = the exact
    answer; bfloat16 = the reference's own run mode (bf16 weights, bf16 latents, bf16 scheduler history - :681,:739)."""
    from oracle.unipc_oracle import UniPCOracle
    p = {k: v.to(dtype) for k, v in dp.items()}
    sch = UniPCOracle()
    sch.set_timesteps(steps, shift=5.0)
    lat = lat0.to(dtype)
    c_, pr, ng, im = cond.to(dtype), prompt.to(dtype), negative.to(dtype), img.to(dtype)
    with torch.no_grad():
        for t in sch.timesteps:
            inp = torch.cat([lat, c_], dim=1)
            ts = t.expand(1)
            c = D.dit_forward(p, dcfg, inp, ts, pr, im)
            u = D.dit_forward(p, dcfg, inp, ts, ng, im)
            c = u + guidance * (c - u)
            lat = sch.step(c.to(lat.dtype), lat)
    return lat.float()


def test_fifty_step_cfg_trajectory_vs_oracle_and_reference_precision():
    """a 50-step, guidance-5 trajectory at narrow width vs the fp32 CPU oracle, with the END-of-trajectory
    error bounded - and the deviation the engine takes from the reference's rounding points quantified: the reference carries
    bf16 latents / scheduler history through the loop (pipeline_chronoedit.py:681,712,739), the engine fp32.  Four runs:
      exact     fp32 oracle                                   (the yardstick)
      ref_bf16  the oracle run entirely in bf16               (what the reference's eager path does; its own error vs exact)
      hip_fp32  the engine, fp32 latents (default)
      hip_bf16  the engine, scheduler.trajectory_dtype = bf16 (latents / history rounded to bf16 after every step)
    Bounds: hip_fp32 vs exact <= 5e-2 and <= the reference's own bf16 error (keeping latents in fp32 must not be WORSE than the
    reference's rounding); hip_bf16 vs ref_bf16 <= 2 x ref_bf16's own error + 2e-2 (same rounding points, different kernels)."""
    from chronoedit_amd.pipeline import denoise
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64, added_kv_proj_dim=256)
    dp = D.make_synthetic_params(dcfg, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(21)
    bf = torch.bfloat16
    lat0 = torch.randn(1, 16, 2, 8, 12, generator=g).to(bf).float()
    cond = torch.randn(1, 20, 2, 8, 12, generator=g).to(bf).float()
    prompt = torch.randn(1, 40, 128, generator=g).to(bf).float()
    negative = torch.randn(1, 40, 128, generator=g).to(bf).float()
    img = torch.randn(1, 257, 64, generator=g).to(bf).float()
    steps, guidance = 50, 5.0
    exact = _oracle_loop(dp, dcfg, lat0, cond, prompt, negative, img, steps, guidance, torch.float32)
    ref_bf16 = _oracle_loop(dp, dcfg, lat0, cond, prompt, negative, img, steps, guidance, bf)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    outs = {}
    for name, tdt in (("hip_fp32", torch.float32), ("hip_bf16", bf)):
        sch = FlowUniPCMultistepScheduler(flow_shift=5.0)
        sch.trajectory_dtype = tdt
        outs[name] = denoise(m, sch, lat0.cuda(), cond.cuda().to(bf), prompt.cuda().to(bf), negative.cuda().to(bf), img.cuda().to(bf),
                             steps, guidance).clone()
        assert torch.isfinite(outs[name]).all()
    e_ref = rel_l2(ref_bf16, exact)
    e_fp32, e_bf16 = rel_l2(outs["hip_fp32"], exact), rel_l2(outs["hip_bf16"], exact)
    d_bf16 = rel_l2(outs["hip_bf16"], ref_bf16)
    print(f"50-step CFG trajectory, end-of-trajectory rel-L2 vs the fp32 oracle: reference-precision (bf16 oracle) {e_ref:.3e} | "
          f"engine fp32 latents {e_fp32:.3e} | engine bf16 trajectory {e_bf16:.3e}; engine bf16 trajectory vs bf16 oracle {d_bf16:.3e}")
    assert e_fp32 < 5e-2 and e_fp32 <= e_ref + 5e-3, (e_fp32, e_ref)
    assert d_bf16 < 2 * e_ref + 2e-2, (d_bf16, e_ref)
