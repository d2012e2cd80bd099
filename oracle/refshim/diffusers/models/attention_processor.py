import torch.nn as nn

from .normalization import RMSNorm


class Attention(nn.Module):
    """Only the configuration the reference instantiates (transformer_chronoedit.py:231-258)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, kv_heads=None, dim_head=64, dropout=0.0,
                 bias=False, qk_norm=None, added_kv_proj_dim=None, added_proj_bias=True, out_bias=True, eps=1e-5,
                 processor=None):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.inner_kv_dim = self.inner_dim if kv_heads is None else dim_head * kv_heads
        self.heads = heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        assert qk_norm == "rms_norm_across_heads"
        self.norm_q = RMSNorm(dim_head * heads, eps=eps)
        self.norm_k = RMSNorm(dim_head * (kv_heads or heads), eps=eps)
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_kv_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_kv_dim, bias=bias)
        self.add_k_proj = self.add_v_proj = None
        self.norm_added_q = self.norm_added_k = None
        if added_kv_proj_dim is not None:
            self.add_k_proj = nn.Linear(added_kv_proj_dim, self.inner_kv_dim, bias=added_proj_bias)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, self.inner_kv_dim, bias=added_proj_bias)
            # Wan: norm across heads on the added k only (no added-q norm)
            self.norm_added_k = RMSNorm(dim_head * (kv_heads or heads), eps=eps)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kwargs)
