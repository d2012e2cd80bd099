"""Golden vectors for the VAE from the REFERENCE's own class definitions.

/root/reference/chronoedit/_src/tokenizers/wan2pt1.py cannot be imported (module-level loguru / easy_io / lazy_config),
but lines 38-581 (CACHE_T ... WanVAE_) use only torch + einops.  This script exec's exactly that source range in a
namespace that provides those imports, loads the seeded synthetic weights of oracle/vae_oracle.make_synthetic_params and
stores encode/decode outputs in tests/golden/vae_*.pt.  Build container only."""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F
from einops import rearrange

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import vae_oracle as V  # noqa: E402

REF = "/root/reference/chronoedit/_src/tokenizers/wan2pt1.py"


def load_reference_classes():
    src = open(REF).read().split("\n")
    start = next(i for i, l in enumerate(src) if l.startswith("CACHE_T = 2"))
    end = next(i for i, l in enumerate(src) if l.startswith("def _video_vae("))
    ns = {"torch": torch, "nn": nn, "F": F, "rearrange": rearrange}
    exec(compile("\n".join(src[start:end]), REF, "exec"), ns)
    return ns


CASES = {
    # name: (cfg kwargs, frames, H, W)
    "small_5f": (dict(dim=32, z_dim=16), 5, 32, 48),
    "small_9f": (dict(dim=32, z_dim=16), 9, 16, 32),
    # the real Wan 2.1 widths (dim 96 -> 96 / 192 / 384 / 384 channels): the MFMA and the HBM-bound conv paths, the Cin padding
    # and the 384-channel mid-block attention at the channel counts the 14B pipeline runs (round-1 VERDICT: toy width only)
    "full_5f": (dict(dim=96, z_dim=16), 5, 64, 96),
    # round 3: a larger frame at the real widths (128 x 192 px: 16 x 24 = 384 mid-block attention tokens, several
    # 128-pixel tiles per conv row) and the temporal-reasoning encode length (29 pixel frames = chunks 1 + 4 x 7 through feat_cache ->
    # 8 latent frames; decode of 8 latent frames frame by frame)
    "full_5f_128x192": (dict(dim=96, z_dim=16), 5, 128, 192),
    "full_29f": (dict(dim=96, z_dim=16), 29, 32, 48),
    # round 4: one frame at 360 x 640 px (latent 45 x 80 = 3 600 mid-block attention tokens; 14 + 26 + 51 M-tiles of
    # 256 pixels per conv layer, rows that end inside a tile, the 96 -> 3 head at 230 400 pixels): the 720p-class row tiling pinned to the
    # REFERENCE's classes, not to the other HIP conv route.  Outputs stored in fp16 (2.8 MB otherwise; 5e-4 against a 5e-2 bound).
    "full_1f_360x640": (dict(dim=96, z_dim=16), 1, 360, 640),
}
HALF = {"full_1f_360x640"}


def main():
    ns = load_reference_classes()
    only = sys.argv[1:]
    for name, (kw, T, H, W) in CASES.items():
        if only and name not in only:
            continue
        cfg = V.VAEConfig(**kw)
        p = V.make_synthetic_params(cfg)
        m = ns["WanVAE_"](dim=cfg.dim, z_dim=cfg.z_dim, dim_mult=list(cfg.dim_mult), num_res_blocks=cfg.num_res_blocks,
                          attn_scales=[], temperal_downsample=list(cfg.temperal_downsample), dropout=0.0).eval()
        sd = m.state_dict()
        assert set(sd) == set(p), (sorted(set(sd) - set(p))[:5], sorted(set(p) - set(sd))[:5])
        m.load_state_dict(p)
        x = torch.rand((1, 3, T, H, W), generator=torch.Generator().manual_seed(5)) * 2 - 1
        zero_scale = [0.0, 1.0]
        with torch.no_grad():
            mu = m.encode(x, zero_scale)
            z = torch.randn(mu.shape, generator=torch.Generator().manual_seed(6))
            rec = m.decode(z, zero_scale)
        extra = {}
        if name in HALF:
            # the error of the reference arithmetic run in ITS eager precision (bf16 weights and activations, the oracle restatement on the
            # host) against this fp32 result - minutes of host time at this size, so it is measured once here and travels with the fixture
            import time
            pb = {k: v.to(torch.bfloat16) for k, v in p.items()}
            t0 = time.perf_counter()
            with torch.no_grad():
                mu_b = V.encode(pb, cfg, x.to(torch.bfloat16)).float()
                rec_b = V.decode(pb, cfg, z.to(torch.bfloat16)).float()
            rl2 = lambda a, b: float((a - b).norm() / b.norm())
            extra = {"bf16_eager_rel_l2": {"mu": rl2(mu_b, mu), "rec": rl2(rec_b, rec), "host_seconds": round(time.perf_counter() - t0, 1)}}
            print(name, "bf16 eager oracle vs fp32 reference:", extra)
            mu, rec = mu.to(torch.float16), rec.to(torch.float16)
        fx = {"cfg": kw, "T": T, "H": H, "W": W, "mu": mu.contiguous(), "rec": rec.contiguous(), **extra,
              "source": "reference wan2pt1.py:38-581 exec'd; weights oracle.vae_oracle.make_synthetic_params(seed 4321)"}
        path = os.path.join(ROOT, "tests", "golden", f"vae_{name}.pt")
        torch.save(fx, path)
        print(name, tuple(mu.shape), tuple(rec.shape), float(mu.abs().mean()), float(rec.abs().mean()), os.path.getsize(path))


if __name__ == "__main__":
    main()
