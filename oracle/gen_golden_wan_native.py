"""Golden vectors for the Wan-native ("sibling stack") front of the DiT: the REFERENCE's own diffsynth model file
(chronoedit_diffsynth/wan_video_dit_chronoedit.py: ``WanModel`` :287-431, blocks :124-285) executed here in fp32 on a tiny
configuration with seeded weights under their NATIVE names.  Pins, together with tests/golden/wan_native_keymap.json:
  * the semantics of the native -> diffusers weight map (chronoedit_amd.weights.wan_native_to_diffusers), not only its names;
  * the diffsynth call path ``model_fn_wan_video(dit, latents=, timestep=, context=, clip_feature=, y=)``
    (wan_video_new_chronoedit.py:1296-1504, the function the diffsynth pipeline actually runs, :95) that
    chronoedit_amd.adapters.model_fn_wan_video mirrors: float timesteps, latents and condition passed apart, CLIP tokens
    first, and PLAIN temporal RoPE positions (``dit.freqs[0][:f]``, :1428-1432).  ``WanModel.forward`` itself (:371-427,
    temporal positions {0, skip_len-1}) cannot run as shipped - it unpacks two values from ``patchify`` (:391), which
    returns one (:356-362) - so it is not executed here;
  * the CPU oracle (oracle/dit_oracle.py) against a SECOND implementation of the same network from the reference.
The two ``diffsynth`` imports of that file are satisfied by stand-in modules (a key-hash helper and a camera adapter the
ChronoEdit configuration never instantiates).  Test infrastructure: runs only in the build container.
    python oracle/gen_golden_wan_native.py   ->  tests/golden/wan_native_tiny.pt"""
import importlib.util
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/chronoedit_diffsynth/wan_video_dit_chronoedit.py"
REF_FN = "/root/reference/chronoedit_diffsynth/wan_video_new_chronoedit.py"

CFG = dict(dim=256, in_dim=36, ffn_dim=512, out_dim=16, text_dim=96, freq_dim=256, eps=1e-6, patch_size=(1, 2, 2), num_heads=2,
           num_layers=2, has_image_input=True, rope_temporal_skip_len=8)


def synth_state_dict(shapes, dim, seed=1234):
    """Seeded weights under the native names, in the fixture's key order (the fixture stores shapes + seed, not tensors)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("modulation"):
            t = torch.randn(shp, generator=g) / dim ** 0.5
        elif len(shp) >= 2:
            t = torch.randn(shp, generator=g) * 0.05
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:  # biases, the LayerNorm weights / biases inside img_emb.proj
            t = 0.05 * torch.randn(shp, generator=g) + (1.0 if k.endswith("weight") else 0.0)
        sd[k] = t
    return sd


def synth_inputs(f, h, w, tlen, text_dim):
    gi = torch.Generator().manual_seed(42 + f)
    x = torch.randn(1, 16, f, h, w, generator=gi)
    y = torch.randn(1, 20, f, h, w, generator=gi)
    ctx = torch.randn(1, tlen, text_dim, generator=gi)
    clip = torch.randn(1, 257, 1280, generator=gi)
    return x, y, ctx, clip


def load_reference_module():
    for name in ("diffsynth", "diffsynth.models", "diffsynth.models.utils", "diffsynth.models.wan_video_camera_controller"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["diffsynth.models.utils"].hash_state_dict_keys = lambda sd, **kw: ""
    sys.modules["diffsynth.models.wan_video_camera_controller"].SimpleAdapter = type("SimpleAdapter", (torch.nn.Module,), {})
    spec = importlib.util.spec_from_file_location("ref_wan_video_dit_chronoedit", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_model_fn(mod):
    """model_fn_wan_video, executed from the reference file's own text (the file as a whole needs the diffsynth package)."""
    from typing import Optional

    from einops import rearrange
    src = open(REF_FN).read()
    start = src.index("def model_fn_wan_video(")
    end = src.index("def model_fn_wans2v(")
    ns = dict(torch=torch, Optional=Optional, rearrange=rearrange, WanModel=mod.WanModel, sinusoidal_embedding_1d=mod.sinusoidal_embedding_1d,
              WanMotionControllerModel=object, VaceWanModel=object, WanAnimateAdapter=object, TeaCache=object,
              TemporalTiler_BCTHW=None, model_fn_wans2v=None)
    exec(src[start:end], ns)
    return ns["model_fn_wan_video"]


def main():
    mod = load_reference_module()
    model_fn = load_reference_model_fn(mod)
    torch.manual_seed(0)
    m = mod.WanModel(**CFG).float().eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth_state_dict(shapes, CFG["dim"])
    m.load_state_dict(sd)
    cases = {}
    for name, (f, h, w), tlen, tval in (("T2", (2, 16, 16), 40, 637.0), ("T8", (8, 8, 12), 77, 21.5)):
        x, y, ctx, clip = synth_inputs(f, h, w, tlen, CFG["text_dim"])
        t = torch.tensor([tval])
        with torch.no_grad():
            out = model_fn(m, latents=x, timestep=t, context=ctx, clip_feature=clip, y=y)
        cases[name] = dict(shape=(f, h, w), text_len=tlen, timestep=t, out=out)
        print(name, tuple(out.shape), float(out.abs().mean()))
    dst = os.path.join(ROOT, "tests", "golden", "wan_native_tiny.pt")
    torch.save(dict(config=CFG, shapes=shapes, weight_seed=1234, cases=cases, source="model_fn_wan_video (wan_video_new_chronoedit.py:1296-1504) over WanModel (wan_video_dit_chronoedit.py:124-431), fp32, CPU"), dst)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
    main()
