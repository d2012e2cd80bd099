"""fp8 mode with the cross-attention on the bf16 two-segment kernel vs under the MXFP8 contract (round 5): error of ONE full-width block at 720p
against the fp32 CPU oracle (the measurement behind tests/test_bench_shapes_gpu.py's bound) and the time of the cross-attention launches.
    python tools/fp8_cross_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402
from chronoedit_amd.transformer import ChronoEditTransformer3DModel  # noqa: E402
from oracle import dit_oracle as O  # noqa: E402

BF = torch.bfloat16
rel = lambda a, b: float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
cfg = O.DiTConfig(num_layers=1)
p_bf = O.make_synthetic_params(cfg, seed=7, dtype=BF)
lat, text, image = O.make_synthetic_inputs(cfg, 2, 90, 160, dtype=BF)
m = ChronoEditTransformer3DModel(num_attention_heads=40, attention_head_dim=128, in_channels=36, out_channels=16, text_dim=4096, freq_dim=256, ffn_dim=13824,
                                 num_layers=1, image_dim=1280, added_kv_proj_dim=5120, device="cuda:0", dtype=BF)
m.load_synthetic_({k: v.cuda() for k, v in p_bf.items()})
ts = torch.tensor([800], device="cuda:0")
args = (torch.cat([lat, lat]).cuda(), torch.cat([ts, ts]), torch.cat([text, text]).cuda(), torch.cat([image, image]).cuda())
outs = {}
outs["bf16"] = m(*args, return_dict=False)[0][:1].float().cpu()
for cross in (False, True):
    m.enable_fp8_gemms().enable_fp8_attention(cross=cross)
    o = m(*args, return_dict=False)[0]
    outs[f"fp8 cross={cross}"] = o[:1].float().cpu()
    with ops.profile() as prof:
        m(*args, return_dict=False)
    s = prof.summary()
    x = {k: round(d["total_ms"], 4) for k, d in s.items() if ("attention" in k and ("512" in k or "257" in k)) or "rmsnorm_rope_mxfp8_14400" in k or "v_mxfp8_transpose_" in k}
    print(f"cross={cross}: cross-attention-side launches (ms): {x}", flush=True)
    tot = sum(d["total_ms"] for d in s.values())
    print(f"cross={cross}: whole forward (1 block, B = 2, nothing cached) {tot:.3f} ms", flush=True)
p32 = {k: v.float() for k, v in p_bf.items()}
with torch.no_grad():
    ref = O.dit_forward(p32, cfg, lat.float(), torch.tensor([800]), text.float(), image.float())
e = {k: rel(v, ref) for k, v in outs.items()}
print({k: f"{v:.3e} ({v / e['bf16']:.2f} x bf16)" for k, v in e.items()})
