"""`scripts/run_inference_diffusers.py::main` ITSELF over the drop-ins (north star: "drops in under scripts/run_inference_diffusers.py").

`oracle/_ref/run_inference_ref.bin` is the reference's script compiled at build time (oracle/build_ref.build_runner: the whole
file - parse_args, calculate_dimensions, main; a marshalled code object, git-ignored, no reference text in the repository).  It is
executed UNMODIFIED with the import names of its header (:64-80) bound to the chronoedit_amd classes (oracle/ref_harness.runner_shims
- exactly the import edit INTEGRATION.md section 1 asks of a maintainer) against a tiny synthetic checkpoint directory in the
diffusers layout (transformer/ vae/ text_encoder/ image_encoder/ tokenizer/ image_processor/ scheduler/), with a real argv:
    CLIPVisionModel / AutoencoderKLWan / ChronoEditTransformer3DModel / ChronoEditPipeline .from_pretrained (:333-364),
    pipe.load_lora_weights + pipe.fuse_lora (:369-376), UniPCMultistepScheduler.from_config(pipe.scheduler.config, flow_shift=) (:379-382),
    pipe.to(device) (:387), calculate_dimensions / --height --width (:403-410), the seeded device generator (:416-418),
    pipe(image=PIL, prompt=str, ...).frames[0] (:428-441), export_to_video + the last-frame PNG (:454-467).
What it writes must equal what chronoedit_amd.pipeline.ChronoEditPipeline.__call__ returns for the same inputs (bit-equal: the same
kernels on the same seeds)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import dit_oracle as D

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNNER = os.path.join(ROOT, "oracle", "_ref", "run_inference_ref.bin")
WORDS = ["make", "the", "cat", "wear", "a", "red", "hat", "dog", "blue", "sky", "turn", "into", "winter", "scene"]


def _write_checkpoint(root):
    """A ChronoEdit model directory in miniature (diffusers layout), every component saved by its own class / library."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import CLIPImageProcessor, PreTrainedTokenizerFast

    from chronoedit_amd import weights
    from chronoedit_amd.clip_vision import CLIPVisionModel
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from chronoedit_amd.umt5 import UMT5EncoderModel
    from chronoedit_amd.vae import wan_vae_param_shapes
    os.makedirs(root)
    dcfg = D.DiTConfig(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=128, image_dim=320, added_kv_proj_dim=256)
    m = ChronoEditTransformer3DModel(num_attention_heads=2, in_channels=36, ffn_dim=512, num_layers=2, text_dim=128, image_dim=320,
                                     added_kv_proj_dim=256, device="cpu")
    m.load_synthetic_(D.make_synthetic_params(dcfg, dtype=torch.bfloat16))
    m.save_pretrained(os.path.join(root, "transformer"))
    os.makedirs(os.path.join(root, "vae"))
    json.dump({"base_dim": 32, "z_dim": 16, "_class_name": "AutoencoderKLWan"}, open(os.path.join(root, "vae", "config.json"), "w"))
    g = torch.Generator().manual_seed(1)
    vae_sd = {}
    for k, s in wan_vae_param_shapes(dim=32, z_dim=16).items():
        fan_in = int(np.prod(s[1:])) if len(s) > 1 else 1
        vae_sd[k] = torch.ones(s) if k.endswith("gamma") else torch.zeros(s) if k.endswith(".bias") else torch.randn(s, generator=g) / fan_in ** 0.5
    save_file(vae_sd, os.path.join(root, "vae", weights.WEIGHTS_NAME))
    torch.manual_seed(2)
    te = UMT5EncoderModel(vocab_size=64, d_model=128, d_kv=64, d_ff=256, num_layers=1, num_heads=2, device="cpu")
    os.makedirs(os.path.join(root, "text_encoder"))
    json.dump({**vars(te.config), "model_type": "umt5"}, open(os.path.join(root, "text_encoder", "config.json"), "w"))
    save_file({k: v.detach().clone() for k, v in te.state_dict().items()}, os.path.join(root, "text_encoder", "model.safetensors"))
    ie = CLIPVisionModel(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14, device="cpu")
    os.makedirs(os.path.join(root, "image_encoder"))
    json.dump(vars(ie.config), open(os.path.join(root, "image_encoder", "config.json"), "w"))
    save_file({k: v.detach().clone() for k, v in ie.state_dict().items()}, os.path.join(root, "image_encoder", "model.safetensors"))
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump({"_class_name": "UniPCMultistepScheduler", "flow_shift": 3.0, "solver_order": 2, "use_flow_sigmas": True,
               "prediction_type": "flow_prediction"}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    # host-side objects of the reference (transformers): a word-level tokenizer with T5's special tokens, a CLIP image processor
    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2, **{w: 3 + i for i, w in enumerate(WORDS)}}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", eos_token="</s>", unk_token="<unk>").save_pretrained(os.path.join(root, "tokenizer"))
    CLIPImageProcessor(size={"shortest_edge": 56}, crop_size={"height": 56, "width": 56}).save_pretrained(os.path.join(root, "image_processor"))
    return m


def _write_lora(path, model):
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(5)
    sd = {}
    for t in ["blocks.0.attn1.to_q", "blocks.0.attn1.to_v", "blocks.1.ffn.net.0.proj", "blocks.1.attn2.to_out.0"]:
        lin = dict(model.named_modules())[t]
        sd[f"transformer.{t}.lora_A.weight"] = torch.randn(8, lin.in_features, generator=g) * 0.05
        sd[f"transformer.{t}.lora_B.weight"] = torch.randn(lin.out_features, 8, generator=g) * 0.05
    save_file(sd, path)


def _run_main(argv, videos):
    """exec the reference runner's code under the import shims and call its main() with `argv`."""
    from oracle import build_ref, ref_harness
    code = build_ref.load_runner()
    old_argv = sys.argv
    with ref_harness.runner_shims(videos):
        ns = {"__name__": "reference_run_inference_diffusers"}
        exec(code, ns)
        sys.argv = ["run_inference_diffusers.py"] + argv
        try:
            ns["main"]()
        finally:
            sys.argv = old_argv
    return ns


@pytest.mark.skipif(not os.path.exists(RUNNER), reason="oracle/_ref/run_inference_ref.bin not built (needs /root/reference at build time)")
@pytest.mark.parametrize("mode", ["plain", "lora", "reasoning", "auto_dims"])
def test_reference_runner_main_over_the_dropins(tmp_path, mode):
    from PIL import Image

    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    root = str(tmp_path / "ChronoEdit-tiny")
    model = _write_checkpoint(root)
    g = torch.Generator().manual_seed(9)
    src = Image.fromarray((torch.rand(80, 120, 3, generator=g) * 255).to(torch.uint8).numpy())
    inp, out_mp4, out_png = str(tmp_path / "in.png"), str(tmp_path / "out" / "edit.mp4"), str(tmp_path / "out" / "edit.png")
    src.save(inp)
    prompt = "make the cat wear a red hat"
    H, W = 64, 96
    argv = ["--model-path", root, "--input", inp, "--output", out_mp4, "--prompt", prompt, "--num-inference-steps", "4", "--guidance-scale", "5.0",
            "--flow-shift", "5.0", "--seed", "7", "--device", "cuda", "--disable-guardrails"]
    if mode != "auto_dims":
        argv += ["--height", str(H), "--width", str(W)]
    lora = str(tmp_path / "distill_lora.safetensors")
    if mode == "lora":
        _write_lora(lora, model)
        argv += ["--lora-path", lora, "--lora-scale", "0.8"]
    if mode == "reasoning":
        argv += ["--enable-temporal-reasoning", "--num-temporal-reasoning-steps", "2"]
    if mode == "auto_dims":  # calculate_dimensions targets 720 x 1280 pixels: keep the test small by running ONE step at that size
        argv[argv.index("--num-inference-steps") + 1] = "1"
    videos = []
    ns = _run_main(argv, videos)
    assert os.path.exists(out_png), "main() did not write the last-frame image"
    got = np.asarray(Image.open(out_png))

    # the same edit through the engine pipeline, built the way main() builds it
    pipe = ChronoEditPipeline.from_pretrained(root, torch_dtype=torch.bfloat16)
    if mode == "lora":
        pipe.load_lora_weights(lora, adapter_name="distill_lora")
        pipe.fuse_lora(adapter_names=["distill_lora"], lora_scale=0.8)
    pipe.scheduler = FlowUniPCMultistepScheduler.from_config(pipe.scheduler.config, flow_shift=5.0)
    pipe.to("cuda")
    if mode == "auto_dims":
        W2, H2 = ns["calculate_dimensions"](src, 16)
        assert (W2 * H2) <= 720 * 1280 and W2 % 16 == 0 and H2 % 16 == 0 and abs(W2 / H2 - 120 / 80) < 0.05
        H, W = H2, W2
    image = src.resize((W, H))
    reasoning = mode == "reasoning"
    frames = pipe(image=image, prompt=prompt, negative_prompt=None, height=H, width=W, num_frames=29 if reasoning else 5,
                  num_inference_steps=1 if mode == "auto_dims" else 4, guidance_scale=5.0, enable_temporal_reasoning=reasoning,
                  num_temporal_reasoning_steps=2 if reasoning else 50, generator=torch.Generator(device="cuda").manual_seed(7),
                  offload_model=False).frames[0]
    want = (frames[-1] * 255).clip(0, 255).astype("uint8")
    assert got.shape == want.shape == (H, W, 3)
    assert np.array_equal(got, want), f"main() frame differs from the pipeline's: max |d| = {np.abs(got.astype(int) - want.astype(int)).max()}"
    assert got.std() > 1.0  # not a constant image
    if reasoning:  # export_to_video(output, args.output, fps=8): 4 reasoning frames + the edited frame
        (path, fps, vid), = videos
        assert path == out_mp4 and fps == 8 and vid.shape == (5, H, W, 3) and np.array_equal(vid, frames)
    else:
        assert not videos
    if mode == "lora":  # the fused adapter is visible in the result
        base = ChronoEditPipeline.from_pretrained(root, torch_dtype=torch.bfloat16)
        base.scheduler = FlowUniPCMultistepScheduler.from_config(base.scheduler.config, flow_shift=5.0)
        plain = base(image=image, prompt=prompt, height=H, width=W, num_frames=5, num_inference_steps=4, guidance_scale=5.0,
                     num_temporal_reasoning_steps=50, generator=torch.Generator(device="cuda").manual_seed(7)).frames[0]
        assert np.abs(plain[-1] - frames[-1]).max() > 1e-3
