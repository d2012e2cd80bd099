"""fp8 mode: the attention kernels with the out-projection's MX operand as their output (ce_attention_mxfp8_quant,
ce_attention_2seg_vt_quant_bf16) against the two-launch form (bf16 output + ce_quant_rows_mxfp8), at the step's shapes, one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16
N, H, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (7200, 40, 2)))
D = H * 128
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * N, 3 * D, generator=g).to(BF).cuda()
one = torch.ones(D).cuda()
q8, sq = ops.rmsnorm_rope_mxfp8(qkv[:, :D], one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
k8, sk = ops.rmsnorm_rope_mxfp8(qkv[:, D:2 * D], one, None, 128, 1e-6)
v8t, sv = ops.v_mxfp8_transpose(qkv[:, 2 * D:], N, B, H)
o = torch.empty((B * N, D), dtype=BF, device="cuda")
o8 = torch.empty((B * N, D), dtype=torch.uint8, device="cuda")
s8 = torch.zeros((ops.mx_scale_bytes(B * N, D),), dtype=torch.uint8, device="cuda")
Tt, Ti = 512, 257
c1, c2 = 512, 264
q = torch.randn(B * N, D, generator=g).to(BF).cuda()
k1 = torch.randn(B * Tt, D, generator=g).to(BF).cuda()
k2 = torch.randn(B * Ti, D, generator=g).to(BF).cuda()
v1t = torch.randn(D, (B - 1) * c1 + 512, generator=g).to(BF).cuda()
v2t = torch.randn(D, (B - 1) * c2 + 320, generator=g).to(BF).cuda()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


cases = {
    "self: attention_mxfp8 -> bf16": lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=o, batch=B),
    "self: quant_rows_mxfp8 of it": lambda: ops.quant_rows_mxfp8(o, out=o8, scale=s8),
    "self: attention_mxfp8 -> MX operand": lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, batch=B, out8=o8, scale8=s8),
    "cross: attention_2seg_vt -> bf16": lambda: ops.attention_2seg_vt(q, k1, v1t, Tt, k2, v2t, Ti, H, out=o, batch=B, cols1=c1, cols2=c2),
    "cross: attention_2seg_vt -> MX operand": lambda: ops.attention_2seg_vt(q, k1, v1t, Tt, k2, v2t, Ti, H, batch=B, cols1=c1, cols2=c2, out8=o8, scale8=s8),
}
for _ in range(2):
    for name, fn in cases.items():
        print(f"{name:42s} {timed(fn) * 1e3:8.1f} us", flush=True)
