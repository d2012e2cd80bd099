"""Sibling-stack fronts on the GPU: chronoedit_amd.adapters over the HIP engine against the output of the
reference's own DiffSynth implementation (tests/golden/wan_native_tiny.pt) and against the diffusers front of this package."""
import os

import pytest
import torch

from oracle.gen_golden_wan_native import synth_inputs, synth_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "wan_native_tiny.pt")
BF = torch.bfloat16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


@pytest.fixture(scope="module")
def dit(gold):
    from chronoedit_amd import adapters as A
    c = gold["config"]
    m = A.WanModel(**c, device="cuda")
    res = m.load_state_dict(synth_state_dict(gold["shapes"], c["dim"], gold["weight_seed"]))
    assert not res.missing_keys and not res.unexpected_keys
    return m


def _inputs(gold, case):
    c = gold["config"]
    f, h, w = gold["cases"][case]["shape"]
    x, y, ctx, clip = synth_inputs(f, h, w, gold["cases"][case]["text_len"], c["text_dim"])
    dev = torch.device("cuda")
    return (x.to(dev, BF), y.to(dev, BF), ctx.to(dev, BF), clip.to(dev, BF), gold["cases"][case]["timestep"].to(dev))


@pytest.mark.parametrize("case", ["T2", "T8"])
def test_model_fn_matches_the_reference_diffsynth_output(gold, dit, case):
    from chronoedit_amd import adapters as A
    x, y, ctx, clip, t = _inputs(gold, case)
    out = A.model_fn_wan_video(dit, latents=x, timestep=t, context=ctx, clip_feature=clip, y=y)
    ref = gold["cases"][case]["out"]
    assert out.shape == ref.shape and out.dtype == BF
    assert rel_l2(out, ref) < 2e-2, rel_l2(out, ref)  # bf16 engine vs the fp32 reference run (the bar of the DiT forward tests)


def test_fronts_agree_with_the_diffusers_front(gold, dit):
    """WanModel.forward / EditWanModel.forward use the temporal positions {0, skip_len-1}: bit-identical to the diffusers front
    of this package on the same weights; model_fn_wan_video (plain positions) differs for two latent frames."""
    from chronoedit_amd import adapters as A
    c = gold["config"]
    x, y, ctx, clip, t = _inputs(gold, "T2")
    a = dit(x, t, ctx, clip_feature=clip, y=y)
    b = dit.transformer(torch.cat([x, y], dim=1), t, ctx, clip, return_dict=False)[0]
    assert torch.equal(a, b)
    plain = A.model_fn_wan_video(dit, latents=x, timestep=t, context=ctx, clip_feature=clip, y=y)
    assert rel_l2(plain, a) > 1e-3
    e = A.EditWanModel(model_type="i2v", in_dim=c["in_dim"], dim=c["dim"], ffn_dim=c["ffn_dim"], freq_dim=c["freq_dim"],
                       text_dim=c["text_dim"], out_dim=c["out_dim"], num_heads=c["num_heads"], num_layers=c["num_layers"],
                       eps=c["eps"], temporal_skip_len=c["rope_temporal_skip_len"], device="cuda")
    e.load_state_dict(dit.state_dict())
    o = e(x, t.reshape(1, 1), ctx, frame_cond_crossattn_emb_B_L_D=clip, y_B_C_T_H_W=y)
    assert torch.equal(o, a)
    with pytest.raises(AssertionError):
        e(x, t, ctx, frame_cond_crossattn_emb_B_L_D=clip, y_B_C_T_H_W=y)  # timesteps must be [B, 1] (wan2pt1.py:780)
    with pytest.raises(NotImplementedError):
        e(x, t.reshape(1, 1), ctx, frame_cond_crossattn_emb_B_L_D=clip, y_B_C_T_H_W=y, slg_layers=[1])


def test_merged_cfg_and_float_timesteps(gold, dit):
    """One latent, two prompts (wan_video_new_chronoedit.py:1399-1404) == two calls; fractional timesteps are honoured."""
    from chronoedit_amd import adapters as A
    x, y, ctx, clip, t = _inputs(gold, "T2")
    ctx2 = torch.cat([ctx, ctx.flip(1)], dim=0)
    both = A.model_fn_wan_video(dit, latents=x, timestep=t, context=ctx2, clip_feature=torch.cat([clip, clip]), y=torch.cat([y, y]))
    one = A.model_fn_wan_video(dit, latents=x, timestep=t, context=ctx2[1:], clip_feature=clip, y=y)
    assert both.shape[0] == 2 and rel_l2(both[1], one[0]) < 2e-3
    frac = A.model_fn_wan_video(dit, latents=x, timestep=t + 0.5, context=ctx, clip_feature=clip, y=y)
    whole = A.model_fn_wan_video(dit, latents=x, timestep=t, context=ctx, clip_feature=clip, y=y)
    assert rel_l2(frac, whole) > 1e-4


def test_native_state_dict_round_trip(gold, dit):
    c = gold["config"]
    native = synth_state_dict(gold["shapes"], c["dim"], gold["weight_seed"])
    sd = dit.state_dict()
    assert set(sd) == set(native)
    for k in ("blocks.1.self_attn.q.weight", "head.modulation", "img_emb.proj.1.weight", "blocks.0.modulation"):
        assert torch.allclose(sd[k].float().cpu(), native[k].to(sd[k].dtype).float()), k
