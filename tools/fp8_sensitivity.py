"""What fp8 costs in accuracy at the FULL width, Linear by Linear: one block of the 14B width (D = 5120, 40 heads, F = 13 824,
512 + 257 context rows) at the 720p shape (N = 7 200) against the fp32 CPU oracle, with

  * everything in bf16 (the yardstick: the reference's own arithmetic),
  * ONE of the six large Linears on the MX fp8 GEMM at a time (attention in bf16),
  * only the self-attention under the MXFP8 contract,
  * the policies of ChronoEditTransformer3DModel.FP8_POLICIES with and without the MXFP8 self-attention,

and - for the step's rate beside each policy - the time of the block's launches (HIP events, ops.profile).  Error of a configuration in quadrature
over the bf16 floor, sqrt(e^2 - e_bf16^2), is what that configuration ADDS; the single-Linear rows add up (in quadrature) to the all-six row.
    python tools/fp8_sensitivity.py [h w]      (default 90 160 = 1280x720; 132 198 = configs[4])"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402
from oracle import dit_oracle as O  # noqa: E402  (the checker: fp32 CPU restatement of the reference's block)

BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(cfg, params):
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    m = ChronoEditTransformer3DModel(
        num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim, in_channels=cfg.in_channels,
        out_channels=cfg.out_channels, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim, ffn_dim=cfg.ffn_dim,
        num_layers=cfg.num_layers, image_dim=cfg.image_dim, added_kv_proj_dim=cfg.added_kv_proj_dim,
        rope_temporal_skip_len=cfg.rope_temporal_skip_len, device="cuda:0", dtype=BF)
    m.load_synthetic_({k: v.to("cuda:0") for k, v in params.items()})
    return m


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (90, 160)
    cfg = O.DiTConfig(num_layers=1)
    p_bf = O.make_synthetic_params(cfg, seed=7, dtype=BF)
    lat, text, image = O.make_synthetic_inputs(cfg, 2, h, w, dtype=BF)
    model = build(cfg, p_bf)
    ts = torch.tensor([800], device="cuda:0")
    args = (lat.cuda(), ts, text.cuda(), image.cuda())
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = O.dit_forward({k: v.float() for k, v in p_bf.items()}, cfg, lat.float(), torch.tensor([800]), text.float(), image.float())
    print(f"# one block, D = 5120, N = {2 * (h // 2) * (w // 2)} tokens ({w * 8}x{h * 8}); fp32 CPU oracle {time.perf_counter() - t0:.1f} s", flush=True)

    names = model.FP8_LINEARS
    configs = [("bf16 (all six Linears + attention)", None, False)]
    configs += [(f"only {n} in fp8", (n,), False) for n in names]
    configs += [("only the self-attention MXFP8", (), True)]
    for pol, lin in model.FP8_POLICIES.items():
        configs += [(f"policy '{pol}' {lin}, attention bf16", lin, False), (f"policy '{pol}' + MXFP8 self-attention", lin, True)]
    extra = [("f1 + f2 (the FFN pair)", ("f1", "f2"), False), ("qkv + q2 (the LayerNorm-fed projections)", ("qkv", "q2"), False),
             ("o1 + o2 (the gated-residual out-projections)", ("o1", "o2"), False), ("all but f2", ("qkv", "o1", "q2", "o2", "f1"), False),
             ("all but o1", ("qkv", "q2", "o2", "f1", "f2"), False), ("all but o2", ("qkv", "o1", "q2", "f1", "f2"), False)]
    configs += extra
    e_bf = None
    rows = []
    for tag, lin, attn8 in configs:
        if lin is None or len(lin) == 0:
            model.enable_fp8_gemms(False)
        else:
            model.enable_fp8_gemms(linears=lin)
        model.enable_fp8_attention(attn8)
        out = model(*args, return_dict=False)[0]
        with ops.profile() as prof:
            out = model(*args, return_dict=False)[0]
        summ = prof.summary()
        blk_ms = sum(d["total_ms"] for k, d in summ.items() if not k.startswith(("gemm_204800", "gemm_1024x", "gemm_514x", "gemm_512x", "gemm_257x")))
        e = rel_l2(out, ref)
        if e_bf is None:
            e_bf = e
        added = max(e * e - e_bf * e_bf, 0.0) ** 0.5
        rows.append((tag, e, e / e_bf, added, blk_ms))
        print(f"{tag:62s} rel-L2 vs fp32 {e:.3e} = {e / e_bf:5.2f} x bf16 | adds {added:.3e} in quadrature | block launches {blk_ms:7.3f} ms", flush=True)
    single = [r for r in rows if r[0].startswith("only ") and "attention" not in r[0]]
    q = sum(r[3] ** 2 for r in single) ** 0.5
    print(f"# quadrature sum of the six single-Linear additions: {q:.3e}  (all six measured: {[r for r in rows if r[0].startswith(chr(112)+'olicy ' + chr(39) + 'fast') and 'bf16' in r[0]][0][3]:.3e})")


if __name__ == "__main__":
    main()
