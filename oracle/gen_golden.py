"""Generate tests/golden/dit_*.pt by executing the REFERENCE's own transformer file.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed and travel to the GPU box.  It imports
/root/reference/chronoedit_diffusers/transformer_chronoedit.py with oracle/refshim on
sys.path (a stand-in for the absent diffusers wheel — leaf modules only), loads the
seeded synthetic state dict of oracle/dit_oracle.make_synthetic_params, runs
``ChronoEditTransformer3DModel.forward`` on the seeded synthetic inputs and stores the
output plus a few intermediate taps.

    python oracle/gen_golden.py            # rewrites tests/golden/dit_*.pt
"""
import importlib.util
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, ROOT)

from oracle import dit_oracle as O  # noqa: E402

REF = "/root/reference/chronoedit_diffusers/transformer_chronoedit.py"

CASES = {
    # name: (cfg kwargs, (T, h, w), text_len, dtype)
    "tiny_T2_fp32": (dict(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=96, image_dim=64, added_kv_proj_dim=256), (2, 16, 16), 40, torch.float32),
    "tiny_T2_bf16": (dict(num_attention_heads=2, ffn_dim=512, num_layers=2, text_dim=96, image_dim=64, added_kv_proj_dim=256), (2, 16, 16), 40, torch.bfloat16),
    "tiny_T8_fp32": (dict(num_attention_heads=2, ffn_dim=512, num_layers=1, text_dim=96, image_dim=64, added_kv_proj_dim=256), (8, 8, 12), 77, torch.float32),
    "tiny_T8_bf16": (dict(num_attention_heads=2, ffn_dim=512, num_layers=1, text_dim=96, image_dim=64, added_kv_proj_dim=256), (8, 8, 12), 77, torch.bfloat16),
    "small_T2_bf16": (dict(num_attention_heads=4, ffn_dim=1536, num_layers=3, text_dim=128, image_dim=96, added_kv_proj_dim=512), (2, 24, 40), 512, torch.bfloat16),
}


def load_reference_module():
    spec = importlib.util.spec_from_file_location("ref_transformer_chronoedit", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_reference_model(mod, cfg: O.DiTConfig, params):
    m = mod.ChronoEditTransformer3DModel(
        patch_size=cfg.patch_size, num_attention_heads=cfg.num_attention_heads,
        attention_head_dim=cfg.attention_head_dim, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
        text_dim=cfg.text_dim, freq_dim=cfg.freq_dim, ffn_dim=cfg.ffn_dim, num_layers=cfg.num_layers,
        cross_attn_norm=cfg.cross_attn_norm, qk_norm=cfg.qk_norm, eps=cfg.eps, image_dim=cfg.image_dim,
        added_kv_proj_dim=cfg.added_kv_proj_dim, rope_max_seq_len=cfg.rope_max_seq_len,
        rope_temporal_skip_len=cfg.rope_temporal_skip_len,
    )
    sd = m.state_dict()
    missing = set(sd) - set(params)
    extra = set(params) - set(sd)
    assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
    # dtype per tensor follows the synthetic dict (fp32 islands kept, transformer_chronoedit.py:338)
    for name, p in m.named_parameters():
        p.data = params[name].clone()
    return m.eval()


def main():
    mod = load_reference_module()
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for name, (kw, (T, h, w), text_len, dtype) in CASES.items():
        cfg = O.DiTConfig(**kw)
        params = O.make_synthetic_params(cfg, seed=1234, dtype=dtype)
        lat, text, image = O.make_synthetic_inputs(cfg, T, h, w, dtype=dtype, text_len=text_len, real_text=min(24, text_len))
        ts = torch.tensor([637], dtype=torch.long)
        model = build_reference_model(mod, cfg, params)
        taps = {}
        hooks = []
        for i, blk in enumerate(model.blocks):
            hooks.append(blk.register_forward_hook(lambda _m, _a, out, i=i: taps.__setitem__(f"blocks.{i}.out", out.detach().clone())))
        with torch.no_grad():
            out = model(lat, ts, text, image, return_dict=False)[0]
        for hk in hooks:
            hk.remove()
        fx = {
            "cfg": kw, "T": T, "h": h, "w": w, "text_len": text_len, "real_text": min(24, text_len),
            "dtype": str(dtype).replace("torch.", ""), "timestep": 637, "param_seed": 1234,
            "out": out.float().contiguous(),
            "taps": {k: v.float()[:, :: max(1, v.shape[1] // 16)].contiguous() for k, v in taps.items()},
            "source": "reference transformer_chronoedit.py executed with oracle/refshim (diffusers stand-in)",
        }
        path = os.path.join(ROOT, "tests", "golden", f"dit_{name}.pt")
        torch.save(fx, path)
        print(name, tuple(out.shape), float(out.float().abs().mean()), os.path.getsize(path))


if __name__ == "__main__":
    main()
