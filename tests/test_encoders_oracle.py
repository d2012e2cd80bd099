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
    # same op sequence in bf16, but bf16 CPU GEMMs round host-dependently (oneDNN blocking follows the ISA): measured 0 on the
    # generating host, 2.8e-3 on another Xeon -> stated tolerance 1e-2 (one bf16 ulp is 7.8e-3 of the value)
    assert rel_l2(hb[-2], fx["penultimate_bf16"]) < 1e-2


def test_umt5_oracle_matches_transformers_golden(golden_dir):
    from oracle import umt5_oracle as U
    fx = torch.load(os.path.join(golden_dir, "umt5_tiny.pt"))
    cfg = U.UMT5Cfg(**fx["cfg"])
    p = U.make_synthetic_params(cfg, fx["param_seed"])
    ids, mask = U.make_synthetic_tokens(cfg, fx["lens"], fx["L"], fx["token_seed"])
    out = U.umt5_encode(p, cfg, ids, mask)
    assert (out - fx["last_fp32"]).abs().max().item() < 2e-5  # padded rows included
    ob = U.umt5_encode({k: v.to(torch.bfloat16) for k, v in p.items()}, cfg, ids, mask)
    valid = mask.bool()
    # transformers 5.x takes the eager softmax in bf16, 4.57.1 (and the oracle) in fp32: both sit ~1e-2 from fp32
    assert rel_l2(ob[valid], fx["last_bf16"][valid]) < 2.5e-2
    assert rel_l2(ob[valid], fx["last_fp32"][valid]) < 2.5e-2
    pe = U.prompt_embeds(out, mask)
    assert pe[0, fx["lens"][0]:].abs().max() == 0 and torch.equal(pe[1], out[1])


def test_relative_position_lut_matches_oracle_buckets():
    """The LUT the HIP softmax indexes with (key - query + L - 1) == the oracle's / transformers' bucket matrix."""
    from chronoedit_amd.umt5 import relative_position_buckets
    from oracle import umt5_oracle as U
    for L in (8, 24, 512):
        lut = relative_position_buckets(L, L, 32, 128)
        pos = torch.arange(L)
        want = U.relative_position_bucket(pos[None, :] - pos[:, None], 32, 128)
        got = lut[(pos[None, :] - pos[:, None]) + L - 1]
        assert torch.equal(got.long(), want)
