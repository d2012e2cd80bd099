"""Oracles of the conditioning encoders vs the golden vectors produced by the REAL transformers classes the reference
instantiates (oracle/gen_golden_clip.py, oracle/gen_golden_umt5.py).  CPU only."""
import os

import pytest
import torch


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_clip_oracle_matches_transformers_golden(golden_dir):
    from oracle import clip_oracle as C
    fx = torch.load(os.path.join(golden_dir, "clip_tiny.pt"))
    cfg = C.CLIPVisionCfg(**fx["cfg"])
    p = C.make_synthetic_params(cfg, fx["param_seed"])
    px = C.make_synthetic_pixels(cfg, fx["batch"], fx["pixel_seed"])
    hs = C.clip_vision_hidden_states(p, cfg, px)
    assert len(hs) == fx["n_hidden_states"] == cfg.num_hidden_layers + 1
    assert torch.equal(hs[0], fx["first_fp32"])
    assert rel_l2(hs[-2], fx["penultimate_fp32"]) < 1e-6
    hb = C.clip_vision_hidden_states({k: v.to(torch.bfloat16) for k, v in p.items()}, cfg, px)
    assert hb[-2].dtype == torch.bfloat16
    assert rel_l2(hb[-2], fx["penultimate_bf16"]) < 2e-3  # same op sequence in bf16; bit-equal on this torch build
