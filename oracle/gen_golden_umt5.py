"""Golden vectors for the UMT5 text encoder: runs the REAL transformers UMT5EncoderModel (the class behind
``pipe.text_encoder``, pipeline_chronoedit.py:205-243) in this container on seeded synthetic weights and token ids and
stores ``last_hidden_state`` in fp32 and bf16.
    python oracle/gen_golden_umt5.py      ->  tests/golden/umt5_tiny.pt
transformers here is 5.15 (the reference pins 4.57.1): in fp32 the two are the same arithmetic; in bf16 the 5.x eager path
takes the softmax in bf16 where 4.57.1 (and the oracle) take it in fp32, which is inside the bf16 tolerance."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import umt5_oracle as U  # noqa: E402


def run_hf(cfg: U.UMT5Cfg, params, ids, mask, dtype):
    from transformers import UMT5Config, UMT5EncoderModel
    hc = UMT5Config(vocab_size=cfg.vocab_size, d_model=cfg.d_model, d_kv=cfg.d_kv, d_ff=cfg.d_ff, num_layers=cfg.num_layers,
                    num_heads=cfg.num_heads, relative_attention_num_buckets=cfg.relative_attention_num_buckets,
                    relative_attention_max_distance=cfg.relative_attention_max_distance, layer_norm_epsilon=cfg.layer_norm_epsilon,
                    feed_forward_proj="gated-gelu", dropout_rate=0.0)
    m = UMT5EncoderModel(hc).eval()
    sd = dict(params)
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "embed_tokens" not in k], res
    m = m.to(dtype)
    with torch.no_grad():
        return m(input_ids=ids, attention_mask=mask).last_hidden_state


def main():
    cfg = U.UMT5Cfg(vocab_size=100, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2)
    params = U.make_synthetic_params(cfg, seed=1357)
    ids, mask = U.make_synthetic_tokens(cfg, lens=[17, 24], L=24, seed=7)
    o32 = run_hf(cfg, params, ids, mask, torch.float32)
    p_bf = {k: v.to(torch.bfloat16).float() for k, v in params.items()}
    obf = run_hf(cfg, p_bf, ids, mask, torch.bfloat16)
    fx = {"cfg": vars(cfg), "param_seed": 1357, "token_seed": 7, "lens": [17, 24], "L": 24, "last_fp32": o32.clone(), "last_bf16": obf.clone(),
          "transformers_version": __import__("transformers").__version__}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "umt5_tiny.pt")
    torch.save(fx, out)
    print("wrote", out, tuple(o32.shape))


if __name__ == "__main__":
    main()
