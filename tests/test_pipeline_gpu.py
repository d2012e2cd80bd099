"""One whole edit (VAE encode -> 4-step CFG denoise -> VAE decode) on the HIP engine vs the fp32 CPU oracle pipeline —
BASELINE.json configs[0] in miniature (small widths, full structure).  Tolerance: final latents rel-L2 <= 6e-2 after 4
steps x 2 forwards x 2 blocks in bf16, decoded video <= 8e-2; also checked: batched CFG == sequential CFG."""
import pytest
import torch

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
    pipe = ChronoEditPipeline(vae, m, FlowUniPCMultistepScheduler(flow_shift=5.0))
    args = (image.cuda().to(bf), prompt.cuda().to(bf), negative.cuda().to(bf), img_emb.cuda().to(bf))
    lat = pipe(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), output_type="latent")
    vid = pipe(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda())
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
    pos, neg = pipe.encode_prompt(ids, am, nids, nam)
    assert pos.shape == neg.shape == (1, 64, 128) and pos.dtype == torch.bfloat16
    assert pos[0, 11:].abs().max().item() == 0 and neg[0, 5:].abs().max().item() == 0 and pos[0, :11].abs().max().item() > 0
    img = pipe.encode_image(torch.randn(1, 3, 56, 56, device="cuda:0"))
    assert img.shape == (1, 17, 320) and torch.isfinite(img.float()).all()
    with pytest.raises(ValueError):
        ChronoEditPipeline(None, None, None).encode_image(torch.zeros(1, 3, 56, 56, device="cuda:0"))


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
    dp32 = {k: v.float() for k, v in dp.items()}
    kw = dict(enable_temporal_reasoning=True, num_temporal_reasoning_steps=2)
    with torch.no_grad():
        lat_ref, vid_ref = P.edit(dp32, dcfg, vp, vcfg, image.to(bf).float(), prompt.to(bf).float(), negative.to(bf).float(),
                                  img_emb.to(bf).float(), lat0.clone(), num_frames=F, steps=4, **kw)
    assert lat_ref.shape[2] == 2 and vid_ref.shape[2] == 5
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=64,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    vae = AutoencoderKLWan({k: v.cuda() for k, v in vp.items()}, dim=32, z_dim=16)
    pipe = ChronoEditPipeline(vae, m, FlowUniPCMultistepScheduler(flow_shift=5.0))
    args = (image.cuda().to(bf), prompt.cuda().to(bf), negative.cuda().to(bf), img_emb.cuda().to(bf))
    lat = pipe(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), output_type="latent", **kw)
    vid = pipe(*args, num_frames=F, num_inference_steps=4, guidance_scale=5.0, latents=lat0.cuda(), **kw)
    e_lat, e_vid = rel_l2(lat, lat_ref), rel_l2(vid, vid_ref)
    print(f"reasoning edit: latents rel-L2 {e_lat:.3e}, video rel-L2 {e_vid:.3e}")
    assert lat.shape == lat_ref.shape and vid.shape == vid_ref.shape
    assert e_lat < 6e-2 and e_vid < 8e-2
    # never truncated (k >= steps): 8 latent frames to the end, reasoning video (all but the last latent frame) + edited frame
    kw2 = dict(enable_temporal_reasoning=True, num_temporal_reasoning_steps=50)
    with torch.no_grad():
        lat_ref2, vid_ref2 = P.edit(dp32, dcfg, vp, vcfg, image.to(bf).float(), prompt.to(bf).float(), negative.to(bf).float(),
                                    img_emb.to(bf).float(), lat0.clone(), num_frames=F, steps=3, **kw2)
    vid2 = pipe(*args, num_frames=F, num_inference_steps=3, guidance_scale=5.0, latents=lat0.cuda(), **kw2)
    print(f"reasoning (no truncation): video rel-L2 {rel_l2(vid2, vid_ref2):.3e}  shape {tuple(vid2.shape)}")
    assert vid2.shape == vid_ref2.shape and rel_l2(vid2, vid_ref2) < 8e-2
