"""Cross-attention of the step (7 200 queries x 2 samples, 512 text + 257 image keys, 40 heads): the register-staged V form the engine
runs (`ce_attention_batched_bf16`) vs the V^T / LDS-DMA form (`ce_attention_2seg_vt_bf16`), one process, interleaved.
python tools/cross_attn_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402


def vt_of(v, B, ln, cols):
    D = v.shape[1]
    vt = torch.zeros((D, B * cols), dtype=v.dtype, device=v.device)
    for b in range(B):
        vt[:, b * cols : b * cols + ln] = v[b * ln : (b + 1) * ln].t()
    return vt


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    H, B, Nq, L1, L2 = 40, 2, 7200, 512, 257
    D = H * 128
    q = torch.randn(B * Nq, D, generator=g).to(torch.bfloat16).to(dev)
    kv1 = torch.randn(B * L1, 2 * D, generator=g).to(torch.bfloat16).to(dev)
    kv2 = torch.randn(B * L2, 2 * D, generator=g).to(torch.bfloat16).to(dev)
    v1t, v2t = vt_of(kv1[:, D:], B, L1, 512), vt_of(kv2[:, D:], B, L2, 320)
    o_a, o_b = torch.empty_like(q), torch.empty_like(q)
    run_a = lambda: ops.attention(q, kv1[:, :D], kv1[:, D:], H, out=o_a, k2=kv2[:, :D], v2=kv2[:, D:], batch=B)
    run_b = lambda: ops.attention_2seg_vt(q, kv1[:, :D], v1t, L1, kv2[:, :D], v2t, L2, H, out=o_b, batch=B)

    def timeit(fn, iters=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ta, tb = [], []
    for _ in range(rounds):
        ta.append(timeit(run_a))
        tb.append(timeit(run_b))
    fl = 4.0 * Nq * (L1 + L2) * 128 * H * B
    d = float((o_a.float() - o_b.float()).norm() / o_a.float().norm())
    print(f"cross-attention {Nq}x({L1}+{L2}) h{H} b{B}: register-staged V best {min(ta):.4f} ms {fl / min(ta) / 1e9:.0f} TF | "
          f"V^T by LDS-DMA best {min(tb):.4f} ms {fl / min(tb) / 1e9:.0f} TF ({(min(ta) / min(tb) - 1) * 100:+.1f} %), rel diff {d:.1e}")


if __name__ == "__main__":
    main()
