"""Blocked-layout vs plain-layout V^T attention at the per-rank shape of an 8-GPU Ulysses group with the guidance pair batched
(5 heads, 2 samples, 29 184 query rows = 8 x 3648, 28 800 valid keys): the layout must cost nothing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chronoedit_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (W, n, N, H, B) in [(8, 3648, 28800, 5, 2), (4, 7232, 28800, 10, 2)]:
    D, T = H * 128, W * n
    plain = torch.randn(B, T, 3 * D, generator=g).to(BF).to(dev)
    blocked = plain.view(B, W, n, 3 * D).permute(1, 0, 2, 3).contiguous().view(W * B * n, 3 * D)
    vt_b = ops.v_transpose_blocked(blocked[:, 2 * D:], H, B, n, N)
    out_b = torch.empty(W * B * n, D, dtype=BF, device=dev)
    # plain reference of the same work: samples stacked, T query rows and T keys each (the padded keys count as work here)
    stacked = plain.reshape(B * T, 3 * D)
    vt_p = ops.v_transpose(stacked[:, 2 * D:], H)
    out_p = torch.empty(B * T, D, dtype=BF, device=dev)
    tb = tp = ttb = 1e9
    for _ in range(3):
        tb = min(tb, timeit(lambda: ops.attention_vt_blocked(blocked[:, :D], blocked[:, D:2 * D], vt_b, H, B, n, N, out=out_b)))
        tp = min(tp, timeit(lambda: ops.attention_vt(stacked[:, :D], stacked[:, D:2 * D], vt_p, H, out=out_p, batch=B)))
        ttb = min(ttb, timeit(lambda: ops.v_transpose_blocked(blocked[:, 2 * D:], H, B, n, N, out=vt_b)))
    fl_b, fl_p = 4.0 * T * N * 128 * H * B, 4.0 * T * T * 128 * H * B
    print(f"W={W} n={n} H={H} B={B}: blocked {tb:.3f} ms {fl_b / tb / 1e9:.0f} TF ({N} valid keys) | plain, same rows, {T} keys {tp:.3f} ms {fl_p / tp / 1e9:.0f} TF "
          f"| blocked V transposer {ttb * 1e3:.1f} us", flush=True)
