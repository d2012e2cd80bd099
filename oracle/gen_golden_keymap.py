"""Golden key map diffusers <-> Wan-native checkpoint names, produced by the REFERENCE's own converter
(chronoedit_diffsynth/wan_video_dit_chronoedit.py:434-505, ``WanModelStateDictConverter.from_diffusers``) executed here on
the key set of a 2-block ChronoEdit transformer with image conditioning.  Both sibling stacks of the reference
(chronoedit_diffsynth ``WanModel`` and chronoedit/_src ``wan2pt1``) use the native names.
    python oracle/gen_golden_keymap.py   ->  tests/golden/wan_native_keymap.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/chronoedit_diffsynth/wan_video_dit_chronoedit.py"


def main():
    import torch
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    src = open(REF).read()
    start = src.index("class WanModelStateDictConverter:")
    end = src.index("    def from_civitai(self, state_dict):")
    ns = {"hash_state_dict_keys": lambda sd: ""}  # the hash only selects a config dict, not the names
    exec(src[start:end], ns)
    conv = ns["WanModelStateDictConverter"]()
    m = ChronoEditTransformer3DModel(num_attention_heads=2, attention_head_dim=128, in_channels=36, out_channels=16, text_dim=64, freq_dim=32,
                                     ffn_dim=256, num_layers=2, image_dim=48, added_kv_proj_dim=256, device="meta")
    keys = [k for k, _ in m.named_parameters()]
    sd = {k: k for k in keys}  # values = the diffusers name, so the converted dict reads native -> diffusers
    native, _ = conv.from_diffusers(sd)
    pairs = sorted((d, n) for n, d in native.items())
    dropped = sorted(set(keys) - {d for d, _ in pairs})
    out = os.path.join(ROOT, "tests", "golden", "wan_native_keymap.json")
    with open(out, "w") as f:
        json.dump({"pairs": pairs, "dropped_by_reference_converter": dropped, "source": "wan_video_dit_chronoedit.py:434-505"}, f, indent=1)
    print("wrote", out, len(pairs), "pairs; dropped:", dropped)


if __name__ == "__main__":
    main()
