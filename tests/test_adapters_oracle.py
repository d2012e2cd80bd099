"""Sibling-stack fronts, CPU side: the oracle and the native -> diffusers weight map against the output of
the reference's OWN DiffSynth implementation (tests/golden/wan_native_tiny.pt: model_fn_wan_video over WanModel, executed by
oracle/gen_golden_wan_native.py).  A second, independently written implementation of the network pins the oracle here."""
import os

import pytest
import torch

from oracle import dit_oracle as O
from oracle.gen_golden_wan_native import synth_inputs, synth_state_dict

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wan_native_tiny.pt")


def _load():
    return torch.load(GOLD, map_location="cpu", weights_only=False)


def _cfg(c, plain):
    return O.DiTConfig(num_attention_heads=c["num_heads"], attention_head_dim=c["dim"] // c["num_heads"], in_channels=c["in_dim"],
                       out_channels=c["out_dim"], text_dim=c["text_dim"], freq_dim=c["freq_dim"], ffn_dim=c["ffn_dim"],
                       num_layers=c["num_layers"], eps=c["eps"], image_dim=1280, added_kv_proj_dim=c["dim"],
                       rope_temporal_skip_len=c["rope_temporal_skip_len"], rope_plain_temporal=plain)


def _diffusers_params(G):
    from chronoedit_amd import weights
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    c = G["config"]
    m = ChronoEditTransformer3DModel(num_attention_heads=c["num_heads"], attention_head_dim=c["dim"] // c["num_heads"],
                                     in_channels=c["in_dim"], out_channels=c["out_dim"], text_dim=c["text_dim"], freq_dim=c["freq_dim"],
                                     ffn_dim=c["ffn_dim"], num_layers=c["num_layers"], image_dim=1280, added_kv_proj_dim=c["dim"],
                                     device="meta")
    native = synth_state_dict(G["shapes"], c["dim"], G["weight_seed"])
    keys = [k for k, _ in m.named_parameters()]
    p = weights.wan_native_to_diffusers(native, keys)
    assert set(p) == set(keys)  # every native tensor found its diffusers name and vice versa
    for k, prm in m.named_parameters():
        assert tuple(p[k].shape) == tuple(prm.shape), k
    return p


@pytest.mark.parametrize("case", ["T2", "T8"])
def test_oracle_reproduces_the_diffsynth_call_path(case):
    G = _load()
    c = G["config"]
    p = _diffusers_params(G)
    f, h, w = G["cases"][case]["shape"]
    x, y, ctx, clip = synth_inputs(f, h, w, G["cases"][case]["text_len"], c["text_dim"])
    out = O.dit_forward(p, _cfg(c, plain=True), torch.cat([x, y], dim=1), G["cases"][case]["timestep"], ctx, clip)
    ref = G["cases"][case]["out"]
    rel = ((out - ref).norm() / ref.norm()).item()
    assert rel < 2e-5, rel  # fp32 vs fp32: summation order and the fp64 / fp32 sinusoid only


def test_temporal_positions_matter_for_two_frames():
    """The same network with the diffusers temporal positions {0, 7} differs from the DiffSynth call path (plain {0, 1}) for two
    latent frames - the adapters must choose per front - and agrees for eight."""
    G = _load()
    c = G["config"]
    p = _diffusers_params(G)
    for case, differs in (("T2", True), ("T8", False)):
        f, h, w = G["cases"][case]["shape"]
        x, y, ctx, clip = synth_inputs(f, h, w, G["cases"][case]["text_len"], c["text_dim"])
        out = O.dit_forward(p, _cfg(c, plain=False), torch.cat([x, y], dim=1), G["cases"][case]["timestep"], ctx, clip)
        ref = G["cases"][case]["out"]
        rel = ((out - ref).norm() / ref.norm()).item()
        assert (rel > 1e-3) == differs, (case, rel)


def test_adapter_signatures_mirror_the_siblings():
    """Keyword-for-keyword the call signatures of the two sibling fronts (no GPU needed to inspect them)."""
    import inspect

    from chronoedit_amd import adapters as A
    fn = list(inspect.signature(A.model_fn_wan_video).parameters)
    assert fn[:9] == ["dit", "motion_controller", "vace", "animate_adapter", "latents", "timestep", "context", "clip_feature", "y"]
    fw = list(inspect.signature(A.WanModel.forward).parameters)
    assert fw[:6] == ["self", "x", "timestep", "context", "clip_feature", "y"]
    ew = list(inspect.signature(A.EditWanModel.forward).parameters)
    assert ew[:7] == ["self", "x_B_C_T_H_W", "timesteps_B_T", "crossattn_emb", "seq_len", "frame_cond_crossattn_emb_B_L_D", "y_B_C_T_H_W"]
    with pytest.raises(NotImplementedError):
        A.WanModel(256, 36, 512, 16, 96, 256, 1e-6, (1, 2, 2), 2, 1, True, add_control_adapter=True, device="meta")
    with pytest.raises(NotImplementedError):
        A.model_fn_wan_video(A.WanModel(256, 36, 512, 16, 96, 256, 1e-6, (1, 2, 2), 2, 1, True, device="meta"), tea_cache=object())
