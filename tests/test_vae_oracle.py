"""The VAE oracle vs golden vectors produced by the reference's own WanVAE_ classes (oracle/gen_golden_vae.py)."""
import os

import pytest
import torch

from oracle import vae_oracle as V


@pytest.mark.parametrize("name", ["small_5f", "small_9f", "full_5f", "full_5f_128x192", "full_29f"])
def test_vae_oracle_matches_reference_golden(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, f"vae_{name}.pt"))
    cfg = V.VAEConfig(**fx["cfg"])
    p = V.make_synthetic_params(cfg)
    x = torch.rand((1, 3, fx["T"], fx["H"], fx["W"]), generator=torch.Generator().manual_seed(5)) * 2 - 1
    with torch.no_grad():
        mu = V.encode(p, cfg, x)
        z = torch.randn(mu.shape, generator=torch.Generator().manual_seed(6))
        rec = V.decode(p, cfg, z)
    assert mu.shape == fx["mu"].shape and rec.shape == fx["rec"].shape
    assert torch.allclose(mu, fx["mu"], atol=1e-5, rtol=1e-5), (mu - fx["mu"]).abs().max()
    assert torch.allclose(rec, fx["rec"], atol=1e-5, rtol=1e-5), (rec - fx["rec"]).abs().max()


def test_vae_14b_config_shapes():
    cfg = V.VAEConfig()
    n = sum(int(torch.tensor(s).prod()) for s in V.param_shapes(cfg).values())
    assert 120e6 < n < 135e6  # Wan 2.1 VAE ~ 127 M parameters
    enc, mid, c = V.encoder_layers(cfg)
    assert c == 384 and [l[0] for l in enc].count("down3d") == 2
