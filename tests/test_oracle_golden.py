"""The CPU oracle must reproduce the fixtures produced by the reference's own
transformer_chronoedit.py (oracle/gen_golden.py) — bit-exact in fp32 and in bf16,
because both execute the same torch CPU kernels in the same order."""
import glob
import os

import pytest
import torch

from oracle import dit_oracle as O


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "dit_*.pt")))


def test_fixtures_present(golden_dir):
    assert len(_cases(golden_dir)) >= 5


@pytest.mark.parametrize("name", ["tiny_T2_fp32", "tiny_T2_bf16", "tiny_T8_fp32", "tiny_T8_bf16", "small_T2_bf16"])
def test_oracle_matches_reference_golden(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, f"dit_{name}.pt"))
    dtype = getattr(torch, fx["dtype"])
    cfg = O.DiTConfig(**fx["cfg"])
    p = O.make_synthetic_params(cfg, seed=fx["param_seed"], dtype=dtype)
    lat, text, image = O.make_synthetic_inputs(cfg, fx["T"], fx["h"], fx["w"], dtype=dtype,
                                               text_len=fx["text_len"], real_text=fx["real_text"])
    taps = {}
    with torch.no_grad():
        out = O.dit_forward(p, cfg, lat, torch.tensor([fx["timestep"]]), text, image, taps=taps)
    assert out.shape == fx["out"].shape
    assert torch.equal(out.float(), fx["out"]), (out.float() - fx["out"]).abs().max()
    for k, v in fx["taps"].items():
        got = taps[k].float()
        got = got[:, :: max(1, got.shape[1] // 16)]
        assert torch.equal(got, v), k


def test_rope_temporal_skip_indices():
    """T=2 uses temporal indices {0, 7} (transformer_chronoedit.py:206-207), T=8 uses 0..7."""
    cfg = O.DiTConfig(num_attention_heads=2, num_layers=1)
    r2 = O.rope_table(cfg, 2, 4, 4)
    r8 = O.rope_table(cfg, 8, 4, 4)
    per = 2 * 2
    assert torch.equal(r2[0, 0, :per], r8[0, 0, :per])
    assert torch.equal(r2[0, 0, per:], r8[0, 0, 7 * per:])
    with pytest.raises(AssertionError):
        O.rope_table(cfg, 5, 4, 4)


def test_flops_match_survey():
    cfg = O.DiTConfig()
    assert abs(O.flops_per_forward(cfg, 7200) / 1e12 - 222.38) < 0.05
    assert abs(O.flops_per_forward(cfg, 28800) / 1e12 - 1389.44) < 0.1
    assert abs(O.flops_per_forward(cfg, 512) / 1e12 - 16.00) < 0.02
