"""One WHOLE edit, measured end to end on one MI355X (BASELINE.json's second metric, sec/edit): token ids + pixel values +
input image in, edited video out, through the drop-in pipeline - UMT5 + CLIP encoders, VAE encode of the conditioning
video, N denoising steps of the 14B DiT (batched CFG + fused flow-UniPC), VAE decode.  Random-init weights of the real
architectures, synthetic inputs (there is no network for checkpoints).  Writes one JSON line.
    python tools/full_edit.py [--steps 50] [--guidance 5.0] [--graph]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (build_model: the 14B architecture with seeded random weights)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--flow-shift", type=float, default=5.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from chronoedit_amd.clip_vision import CLIPVisionModel
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.umt5 import UMT5EncoderModel
    from chronoedit_amd.vae import AutoencoderKLWan

    t_build = time.perf_counter()
    torch.manual_seed(0)
    pipe = ChronoEditPipeline(vae=AutoencoderKLWan.random_init(dev, seed=4321),
                              transformer=bench.build_model(40, dev), scheduler=FlowUniPCMultistepScheduler(flow_shift=a.flow_shift),
                              text_encoder=UMT5EncoderModel(device=dev), image_encoder=CLIPVisionModel(device=dev))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build

    g = torch.Generator(device=dev).manual_seed(42)
    image = torch.rand((1, 3, a.height, a.width), generator=g, device=dev) * 2 - 1
    ids = torch.randint(2, 256384, (1, 512), generator=g, device=dev)
    am = torch.zeros((1, 512), dtype=torch.long, device=dev)
    am[0, :64] = 1
    nids = torch.randint(2, 256384, (1, 512), generator=g, device=dev)
    nam = torch.zeros((1, 512), dtype=torch.long, device=dev)
    nam[0, :20] = 1
    px = torch.randn((1, 3, 224, 224), generator=g, device=dev)

    def edit(steps):
        pos, neg = pipe.encode_prompt(input_ids=ids, attention_mask=am, negative_input_ids=nids, negative_attention_mask=nam)
        img = pipe.encode_image(px)
        return pipe.edit_tensors(image, pos, neg if a.guidance > 1 else None, img, num_frames=a.frames, num_inference_steps=steps,
                                 guidance_scale=a.guidance)

    edit(2)  # warm-up: packs every engine, sizes the workspaces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    video = edit(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "sec/edit", "value": round(dt, 3), "unit": "s per edit (1 x MI355X)", "higher_is_better": False, "dtype": "bf16",
           "data": "synthetic (random-init weights of the real architectures, random inputs)",
           "config": {"workload": f"ChronoEdit-14B edit, {a.width}x{a.height}, {a.frames} frames, {a.steps} steps, guidance {a.guidance}: "
                                  "UMT5-XXL x2 prompts + CLIP ViT-H + VAE encode + denoising loop + VAE decode"},
           "steps": a.steps, "sec_per_step_incl_everything": round(dt / a.steps, 4), "output_shape": list(video.shape),
           "finite": bool(torch.isfinite(video.float()).all().item()), "model_build_s": round(t_build, 1),
           "hbm_allocated_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
