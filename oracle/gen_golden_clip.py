"""Golden vectors for the CLIP image encoder: runs the REAL transformers CLIPVisionModel (the class the reference
instantiates, run_inference_diffusers.py:333-338) in this container on seeded synthetic weights and pixels and stores
``hidden_states[-2]`` - what pipeline_chronoedit.py:247-256 feeds the transformer - in fp32 and bf16.
    python oracle/gen_golden_clip.py      ->  tests/golden/clip_tiny.pt
transformers here is 5.15 (the reference pins 4.57.1); the vision tower's arithmetic is unchanged between the two."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import clip_oracle as C  # noqa: E402


def run_hf(cfg: C.CLIPVisionCfg, params, pixels, dtype):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    hc = CLIPVisionConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                          num_attention_heads=cfg.num_attention_heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
                          num_channels=cfg.num_channels, layer_norm_eps=cfg.layer_norm_eps, hidden_act=cfg.hidden_act)
    m = CLIPVisionModel(hc).eval()
    own = dict(m.state_dict())
    if not any(k.startswith("vision_model.") for k in own):  # transformers >= 5 dropped the prefix the 4.x checkpoints carry
        params = {k[len("vision_model."):]: v for k, v in params.items()}
    sd = {k: v for k, v in params.items() if k in own}
    missing = set(k for k in own if "position_ids" not in k) - set(sd)
    assert not missing, missing
    m.load_state_dict(sd, strict=False)
    m = m.to(dtype)
    with torch.no_grad():
        out = m(pixel_values=pixels.to(dtype), output_hidden_states=True)
    return out.hidden_states


def main():
    cfg = C.CLIPVisionCfg(hidden_size=320, intermediate_size=640, num_hidden_layers=3, num_attention_heads=4, image_size=56, patch_size=14)
    params = C.make_synthetic_params(cfg, seed=2468)
    pixels = C.make_synthetic_pixels(cfg, batch=2, seed=11)
    hs32 = run_hf(cfg, params, pixels, torch.float32)
    p_bf = {k: v.to(torch.bfloat16) for k, v in params.items()}
    hs_bf = run_hf(cfg, {k: v.float() for k, v in p_bf.items()}, pixels, torch.bfloat16)
    fx = {"cfg": vars(cfg), "param_seed": 2468, "pixel_seed": 11, "batch": 2,
          "penultimate_fp32": hs32[-2].clone(), "penultimate_bf16": hs_bf[-2].clone(), "n_hidden_states": len(hs32),
          "first_fp32": hs32[0].clone(), "transformers_version": __import__("transformers").__version__}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "clip_tiny.pt")
    torch.save(fx, out)
    print("wrote", out, {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in fx.items() if k != "cfg"})


if __name__ == "__main__":
    main()
