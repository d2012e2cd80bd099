"""Fixed cost per launch of the attention kernels: Nq = 7200 queries x 40 heads x 2 samples against Nkv = 64 ... 7200 keys, bf16 (V^T / LDS-DMA
form) and MXFP8.  time(Nkv) = a + b * tiles: `a` is what a work item pays outside its key loop (Q fetch, first tiles' HBM latency, epilogue),
spread over 9.06 items per CU.   python tools/attn_overhead_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
H, B, Nq = 40, 2, 7200
D = H * 128
BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best


g = torch.Generator().manual_seed(0)
q = torch.randn(B * Nq, D, generator=g).to(BF).to(dev)
one = torch.ones(D, device=dev)
q8, sq = ops.rmsnorm_rope_mxfp8(q, one, None, 128, 1e-6, post_scale=ops.MXFP8_Q_SCALE)
out = torch.empty(B * Nq, D, dtype=BF, device=dev)
rows = []
for Nkv in (64, 128, 256, 512, 1024, 2048, 4096, 7200):
    kv = torch.randn(B * Nkv, 2 * D, generator=g).to(BF).to(dev)
    k8, sk = ops.rmsnorm_rope_mxfp8(kv[:, :D], one, None, 128, 1e-6)
    v8t, sv = ops.v_mxfp8_transpose(kv[:, D:], Nkv, B, H)
    t8 = timeit(lambda: ops.attention_mxfp8(q8, sq, k8, sk, v8t, sv, H, out=out, batch=B))
    t16 = timeit(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H, out=out, batch=B))  # (register-staged bf16 kernel: any key count)
    rows.append((Nkv, (Nkv + 63) // 64, t16, t8))
    print(f"Nkv={Nkv:5d} tiles={(Nkv + 63) // 64:4d}: bf16 {t16:.4f} ms | mxfp8 {t8:.4f} ms", flush=True)
    del kv, k8, sk, v8t, sv
for name, col in (("bf16", 2), ("mxfp8", 3)):
    (n0, t0), (n1, t1) = [(r[1], r[col]) for r in rows if r[0] in (2048, 7200)]
    b = (t1 - t0) / (n1 - n0)
    a = t1 - b * n1
    print(f"{name}: per 64-key tile {b * 1e3:.2f} us, fixed {a * 1e3:.1f} us per launch = {a / t1 * 100:.1f} % of the 7200-key launch "
          f"({a * 1e3 / 9.06:.1f} us per work item and CU)")
