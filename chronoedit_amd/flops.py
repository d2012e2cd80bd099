"""Algorithmic work of the hot path, for the bench's roofline figures.  Product-side arithmetic on shapes only."""


def dit_flops_per_forward(num_tokens: int, *, dim: int = 5120, ffn_dim: int = 13824, num_layers: int = 40, text_len: int = 512,
                          image_len: int = 257, patch_in: int = 144, patch_out: int = 64, has_image: bool = True) -> float:
    """F_fwd(N) = L [ 8 N D^2 (self q,k,v,o) + 4 N^2 D (self QK^T + PV) + 4 N D^2 (cross q,o) + 4 (Tt+Ti) D^2 (cross k,v)
    + 4 N (Tt+Ti) D (cross QK^T + PV) + 4 N D F (FFN) ] + 2 N patch_in D (patch embedding) + 2 N D patch_out (head);
    the uncached count (the step-invariant context projections are included every forward).
    ChronoEdit-14B: 16.00 TFLOP at N = 512, 222.38 at 7 200, 463.81 at 13 068, 1 389.44 at 28 800."""
    N, D, F, L = num_tokens, dim, ffn_dim, num_layers
    ctx = text_len + (image_len if has_image else 0)
    per_layer = 8 * N * D * D + 4 * N * N * D + 4 * N * D * D + 4 * ctx * D * D + 4 * N * ctx * D + 4 * N * D * F
    return float(L * per_layer + 2 * N * patch_in * D + 2 * N * D * patch_out)
