"""The drop-in boundary, proven with the reference's OWN caller code: `oracle/_ref/pipeline_ref.bin` holds the compiled code of
ChronoEditPipeline.__call__ / prepare_latents / encode_prompt / encode_image / check_inputs, lifted verbatim from
/root/reference/chronoedit_diffusers/pipeline_chronoedit.py and compiled by oracle/build_ref.py at build time (a marshalled code
object: git-ignored, shipped to the GPU box like the built .so; no reference source text lives in this repository).  Here it RUNS - `self.transformer(hidden_states=..., timestep=..., encoder_hidden_states=...,
encoder_hidden_states_image=..., attention_kwargs=..., return_dict=False)[0]`, `self.scheduler.step(noise_pred, t, latents,
return_dict=False)[0]`, the in-place slicing of `scheduler.model_outputs` / `last_sample`, `self.vae.encode / decode`,
`self.image_encoder(**image, output_hidden_states=True)` - over the chronoedit_amd drop-ins, and its frames are compared with
chronoedit_amd.pipeline.ChronoEditPipeline.__call__ given the same keyword arguments (the call of
scripts/run_inference_diffusers.py:428-441).

Tolerances (bf16 engine on both sides, 4 steps x 2 forwards x 2 blocks): with `scheduler.trajectory_dtype = bfloat16` (the
reference's rounding points: bf16 latents and history) frames rel-L2 <= 1e-2; with the engine's default fp32 trajectory <= 3e-2
(the deviation of keeping latents in fp32 is thereby bounded against the reference's own loop)."""
import os

import pytest
import torch

from oracle import dit_oracle as D
from oracle import vae_oracle as V

pytestmark = pytest.mark.gpu

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "pipeline_ref.bin")


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


class _ImageProcessor:
    """CLIPImageProcessor stand-in (host-side transformers object in the reference): resize to the encoder's input, normalise."""

    def __init__(self, size):
        self.size = size

    def __call__(self, images=None, return_tensors="pt"):
        from chronoedit_amd.pipeline import ChronoEditPipeline
        px = ChronoEditPipeline.preprocess_image(images, self.size, self.size)

        class _Batch(dict):
            def to(self, device):
                return _Batch({k: v.to(device) for k, v in self.items()})
        return _Batch(pixel_values=px)


def _components():
    from chronoedit_amd.clip_vision import CLIPVisionModel
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from chronoedit_amd.vae import AutoencoderKLWan
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=320, added_kv_proj_dim=256)
    dp = D.make_synthetic_params(dcfg, dtype=torch.bfloat16)
    vp = V.make_synthetic_params(V.VAEConfig(dim=32, z_dim=16))
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=320,
                                     added_kv_proj_dim=256, device="cuda:0")
    m.load_synthetic_({k: v.cuda() for k, v in dp.items()})
    vae = AutoencoderKLWan({k: v.cuda() for k, v in vp.items()}, dim=32, z_dim=16)
    torch.manual_seed(0)
    ie = CLIPVisionModel(hidden_size=320, intermediate_size=640, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14,
                         device="cuda:0")
    mk_sched = lambda: FlowUniPCMultistepScheduler(flow_shift=5.0, sigma_grid="diffusers")
    return m, vae, ie, mk_sched


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/pipeline_ref.bin not built (needs /root/reference at build time)")
@pytest.mark.parametrize("reasoning", [False, True])
def test_reference_call_source_over_the_dropins_matches_the_engine_pipeline(reasoning):
    from PIL import Image

    from chronoedit_amd.pipeline import ChronoEditPipeline, WanPipelineOutput
    from oracle import build_ref
    RefChronoEditPipeline = build_ref.load().RefChronoEditPipeline
    m, vae, ie, mk_sched = _components()
    g = torch.Generator().manual_seed(3)
    H, W = 64, 96
    F = 29 if reasoning else 5
    T = (F - 1) // 4 + 1
    image = Image.fromarray((torch.rand(80, 120, 3, generator=g) * 255).to(torch.uint8).numpy())  # resized by the pipeline
    prompt = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    negative = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    lat0 = torch.randn(1, 16, T, H // 8, W // 8, generator=g)
    kw = dict(image=image, prompt_embeds=prompt, negative_prompt_embeds=negative, height=H, width=W, num_frames=F,
              num_inference_steps=4, guidance_scale=5.0, latents=lat0.clone(), enable_temporal_reasoning=reasoning,
              num_temporal_reasoning_steps=2 if reasoning else 0, offload_model=False)
    proc = _ImageProcessor(56)

    ref = RefChronoEditPipeline(image_encoder=ie, image_processor=proc, transformer=m, vae=vae, scheduler=mk_sched())
    ref.scheduler.trajectory_dtype = torch.bfloat16
    out_ref = ref(**kw)
    frames_ref = out_ref.frames[0]
    assert frames_ref.shape == (5, H, W, 3)  # reasoning mode: 4 reasoning frames + the edited frame (pipeline_chronoedit.py:776-779)

    pipe = ChronoEditPipeline(image_encoder=ie, image_processor=proc, transformer=m, vae=vae, scheduler=mk_sched())
    pipe.scheduler.trajectory_dtype = torch.bfloat16
    out = pipe(**kw)
    assert isinstance(out, WanPipelineOutput)
    frames = out.frames[0]
    assert frames.shape == frames_ref.shape and frames.dtype == frames_ref.dtype
    e_bf16 = rel_l2(frames, frames_ref)
    pipe.scheduler.trajectory_dtype = torch.float32
    e_fp32 = rel_l2(pipe(**kw).frames[0], frames_ref)
    lat_ref = ref(**dict(kw, output_type="latent")).frames
    pipe.scheduler.trajectory_dtype = torch.bfloat16
    lat = pipe(**dict(kw, output_type="latent")).frames
    e_lat = rel_l2(lat, lat_ref)
    print(f"reference __call__ source over drop-ins vs engine pipeline (reasoning={reasoning}): frames rel-L2 {e_bf16:.3e} "
          f"(bf16 trajectory), {e_fp32:.3e} (fp32 trajectory); final latents {e_lat:.3e} (bit-equal: {torch.equal(lat.float().cpu(), lat_ref.float().cpu())})")
    assert e_bf16 < 1e-2 and e_lat < 1e-2 and e_fp32 < 3e-2
    # tuple return, as the reference
    tup = pipe(**dict(kw, return_dict=False))
    assert isinstance(tup, tuple) and len(tup) == 1


def test_engine_pipeline_call_contract():
    """Reference argument checks and hooks on the engine pipeline itself (no lifted source needed): error messages of
    check_inputs (pipeline_chronoedit.py:332-390), the callback protocol (:741-749), `.frames` / output types, interrupt."""
    from chronoedit_amd.pipeline import ChronoEditPipeline
    m, vae, ie, mk_sched = _components()
    pipe = ChronoEditPipeline(image_encoder=ie, image_processor=_ImageProcessor(56), transformer=m, vae=vae, scheduler=mk_sched())
    g = torch.Generator().manual_seed(4)
    img = torch.rand(1, 3, 64, 96, generator=g)
    pe = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    ne = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    base = dict(image=img, prompt_embeds=pe, negative_prompt_embeds=ne, height=64, width=96, num_frames=5, num_inference_steps=3)
    with pytest.raises(ValueError, match="divisible by 16"):
        pipe(**dict(base, height=60))
    with pytest.raises(ValueError, match="Provide either `prompt` or `prompt_embeds`"):
        pipe(image=img, height=64, width=96)
    with pytest.raises(ValueError, match="Cannot forward both `prompt`"):
        pipe(**dict(base, prompt="x"))
    with pytest.raises(ValueError, match="tokenizer"):
        pipe(image=img, prompt="a cat", height=64, width=96)
    with pytest.raises(ValueError, match="callback_on_step_end_tensor_inputs"):
        pipe(**dict(base, callback_on_step_end=lambda *a: {}, callback_on_step_end_tensor_inputs=["nope"]))
    seen = []

    def cb(p, i, t, kwargs):
        seen.append((i, int(t), sorted(kwargs)))
        if i == 1:
            p._interrupt = True  # the reference's `pipe.interrupt` protocol: remaining steps are skipped
        return {}

    out = pipe(**dict(base, callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"], output_type="pt"))
    assert [s[0] for s in seen] == [0, 1] and seen[0][2] == ["latents", "prompt_embeds"]
    assert out.frames.shape == (1, 5, 3, 64, 96) and float(out.frames.min()) >= 0 and float(out.frames.max()) <= 1
    pil = pipe(**dict(base, output_type="pil")).frames
    assert len(pil) == 1 and len(pil[0]) == 5 and pil[0][0].size == (96, 64)
    # seeded generator on the device, as run_inference_diffusers.py:416 builds it: reproducible
    a = pipe(**dict(base, generator=torch.Generator(device="cuda:0").manual_seed(7), output_type="latent")).frames
    b = pipe(**dict(base, generator=torch.Generator(device="cuda:0").manual_seed(7), output_type="latent")).frames
    assert torch.equal(a, b)


def test_engine_pipeline_batch_of_edits_and_callback_returned_embeds():
    """batch_size * num_videos_per_prompt > 1 (pipeline_chronoedit.py:493,631-637,676-691): the edits run one after the other on the
    engine and every sample equals its own single call; a callback that returns `prompt_embeds` / `negative_prompt_embeds` replaces
    the conditioning from the next step on (:747-749)."""
    from chronoedit_amd.pipeline import ChronoEditPipeline
    m, vae, ie, mk_sched = _components()
    pipe = ChronoEditPipeline(image_encoder=ie, image_processor=_ImageProcessor(56), transformer=m, vae=vae, scheduler=mk_sched())
    g = torch.Generator().manual_seed(6)
    img = torch.rand(1, 3, 64, 96, generator=g)
    pe = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    ne = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    lat = torch.randn(2, 16, 2, 8, 12, generator=g)
    base = dict(image=img, prompt_embeds=pe, negative_prompt_embeds=ne, height=64, width=96, num_frames=5, num_inference_steps=3, output_type="pt")
    both = pipe(**dict(base, prompt_embeds=pe.repeat(2, 1, 1), negative_prompt_embeds=ne.repeat(2, 1, 1), latents=lat.clone())).frames  # batch_size = 2
    assert both.shape == (2, 5, 3, 64, 96)
    for b in range(2):
        one = pipe(**dict(base, latents=lat[b:b + 1].clone())).frames
        assert torch.equal(both[b:b + 1], one), b
    assert not torch.equal(both[0], both[1])
    other = torch.randn(1, 40, 128, generator=g).to(torch.bfloat16).cuda()
    seen = []

    def cb(p, i, t, kwargs):
        seen.append(i)
        return {"prompt_embeds": other} if i == 0 else {}

    swapped = pipe(**dict(base, latents=lat[:1].clone(), callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"])).frames
    plain = pipe(**dict(base, latents=lat[:1].clone())).frames
    assert seen == [0, 1, 2] and not torch.equal(swapped, plain)
    # ... and equals running step 0 with `pe`, the rest with `other`: the engine honours the returned tensor
    from chronoedit_amd.pipeline import denoise, prepare_latents, decode_latents
    pipe.use_graph = False
    eager = pipe(**dict(base, latents=lat[:1].clone(), callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"])).frames
    assert torch.equal(eager, swapped)  # hipGraph replay (default) == eager loop, also across a conditioning swap
    with pytest.raises(ValueError, match="bfloat16"):
        pipe.to("cuda", dtype=torch.float32)
